from .matrix_factorization import MF
