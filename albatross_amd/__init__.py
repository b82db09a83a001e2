"""MI355X-native dense Gaussian-process fit/predict engine (albatross hot path).

Host-side mirror of albatross's GaussianProcessRegression / CovarianceFunction
call surface over the C-ABI in include/albatross_amd.h.  All Gram / factor /
solve / predict arithmetic runs in the HIP library; there is no CPU fallback.
"""
from .covariance import (AngularDistance, Constant, CovarianceFunction, EuclideanDistance, Exponential,
                         FeatureSet, IndependentNoise, LinearCombination, Matern32, Matern52, Measurement, MeasurementOnly,
                         Nugget, Polynomial, ProductOfCovarianceFunctions, RadialDistance, ScalingFunction,
                         ScalingTerm, SquaredExponential, SumOfCovarianceFunctions, as_measurements,
                         measurement_only, OnlyForAlternatives, VariantFeatures, only_for_alternatives)

from .gp import (AlbatrossAmdError, BlockSymmetric, ExplainedCovariance, PivotedLDLT, Context, DeviceArray, CrossValidation, CrossValidationPrediction, DenseFactor,
                 LeaveOneOutGrouper, group_indexer, root_mean_square_error, UpdatedGPFit, negative_log_likelihood, FitModel, GaussianProcessRegression, GPFit, JointDistribution,
                 LinearMean, MeanFunction, SumOfMeanFunctions, ProductOfMeanFunctions, MarginalDistribution, NanInputError, NotPositiveDefiniteError, Prediction,
                 RegressionDataset, ZeroMean, default_context, fit_batch, gp_from_covariance, gp_from_covariance_and_mean)

from .sparse_gp import (FixedInducingPoints, SparseFitModel, SparseGaussianProcessRegression, SparseGPFit,
                        UniformlySpacedInducingPoints, rebase_inducing_points, sparse_gp_from_covariance,
                        sparse_gp_from_covariance_and_mean)

__all__ = [n for n in dir() if not n.startswith("_")]
