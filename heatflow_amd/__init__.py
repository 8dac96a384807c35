"""heatflow_amd: MI355X-native transient axisymmetric heat solver (hot path of cebarker1000/heatflow)."""
__version__ = "0.1.0"
