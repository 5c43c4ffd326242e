from .gmm import MixtureOfGaussians, BayesianMixtureOfGaussians
from .ilr import MixtureOfLinearGaussians, BayesianMixtureOfLinearGaussians
from .hgmm import (BayesianMixtureOfGaussiansWithHierarchicalPrior, MixtureOfMixtureOfGaussians,
                   BayesianMixtureOfMixtureOfGaussians)
from .hilr import (BayesianMixtureOfLinearGaussiansWithTiedActivation, MixtureOfMixtureOfLinearGaussians,
                   BayesianMixtureOfMixtureOfLinearGaussians)
