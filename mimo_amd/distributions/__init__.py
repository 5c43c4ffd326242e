from .gating import Categorical, Dirichlet, TruncatedStickBreaking
from .wishart import Wishart
from .gaussian import (StackedGaussiansWithPrecision, TiedGaussiansWithPrecision,
                       StackedGaussiansWithDiagonalPrecision, TiedGaussiansWithDiagonalPrecision)
from .lingauss import StackedLinearGaussiansWithPrecision, TiedLinearGaussiansWithPrecision
from .composite import (StackedNormalWisharts, StackedMatrixNormalWisharts, TiedNormalWisharts,
                        TiedMatrixNormalWisharts, StackedNormalGammas, TiedNormalGammas)
from .bayesian import (CategoricalWithDirichlet, CategoricalWithStickBreaking,
                       StackedGaussiansWithNormalWisharts, StackedLinearGaussiansWithMatrixNormalWisharts,
                       TiedGaussiansWithNormalWisharts, TiedLinearGaussiansWithMatrixNormalWisharts,
                       StackedGaussiansWithNormalGammas, TiedGaussiansWithNormalGammas)
from .hierarchical import (NormalWishart, TiedGaussiansWithScaledPrecision,
                           TiedGaussiansWithHierarchicalNormalWisharts, MatrixNormalWithPrecision,
                           StackedAffineLinearGaussiansWithPrecision,
                           TiedAffineLinearGaussiansWithMatrixNormalWisharts)
