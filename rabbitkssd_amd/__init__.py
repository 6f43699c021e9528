"""rabbitkssd_amd: MI355X-native sketch + distance engine behind RabbitKSSD's hot path.

The product is the C-ABI library (include/rabbitkssd.h, rabbitkssd_amd/csrc) and the C++
host tool built on it; this Python package only holds the ctypes binding used by tests and
bench.py, the build helper and the synthetic-input generators."""
__version__ = "0.1.0"
