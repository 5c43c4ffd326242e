"""mimo_amd — MI355X-native E-step / sufficient-statistics engine behind the mimo API."""
