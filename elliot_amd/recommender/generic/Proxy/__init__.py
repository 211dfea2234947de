from .Proxy import ProxyRecommender
