"""Host-side mirror of Elliot's plugin surface for the latent-factor path."""
from .base_recommender_model import BaseRecommenderModel, init_charger
from .recommender_utils_mixin import RecMixin
from .latent_factor_models.BPRMF_batch.BPRMF_batch import BPRMF_batch
from .latent_factor_models.BPRMF.BPRMF import BPRMF
from .latent_factor_models.MF.matrix_factorization import MF
from .latent_factor_models.PMF.probabilistic_matrix_factorization import PMF
from .latent_factor_models.FunkSVD.funk_svd import FunkSVD
from .latent_factor_models.LogisticMF.logistic_matrix_factorization import LMF, LogisticMatrixFactorization
from .latent_factor_models.CML.CML import CML
from .latent_factor_models.MF2020.MF import MF2020
from .graph_based.lightgcn.LightGCN import LightGCN
from .graph_based.ngcf.NGCF import NGCF
from .generic.Proxy.Proxy import ProxyRecommender
from .autoencoders.vae.multi_vae import MultiVAE
from .autoencoders.dae.multi_dae import MultiDAE
from .neural.NeuMF.neural_matrix_factorization import NeuMF
from .neural.GeneralizedMF.generalized_matrix_factorization import GMF

__all__ = ["BaseRecommenderModel", "init_charger", "RecMixin", "BPRMF_batch", "BPRMF", "MultiVAE", "MultiDAE", "NeuMF", "GMF",
           "MF", "PMF", "FunkSVD", "LogisticMatrixFactorization", "LMF", "CML", "MF2020", "LightGCN", "NGCF", "ProxyRecommender"]
