from .gating import Categorical, Dirichlet, TruncatedStickBreaking
from .wishart import Wishart
from .gaussian import StackedGaussiansWithPrecision
from .lingauss import StackedLinearGaussiansWithPrecision
from .composite import StackedNormalWisharts, StackedMatrixNormalWisharts
from .bayesian import (CategoricalWithDirichlet, CategoricalWithStickBreaking,
                       StackedGaussiansWithNormalWisharts, StackedLinearGaussiansWithMatrixNormalWisharts)
