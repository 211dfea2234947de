from .CML import CML
