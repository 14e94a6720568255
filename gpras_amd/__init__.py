"""gpras_amd: MI355X-native GP regression engine behind gpras's GPRAS.fit / predict surface."""

__version__ = "0.1.0"
