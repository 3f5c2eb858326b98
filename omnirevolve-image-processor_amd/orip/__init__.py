"""orip -- Python host side of the MI355X-native hot path of omnirevolve-image-processor.

Mirrors the reference's stage interface (02_color_extract ... 12_optimize_plot_order, pipeline.py) on top of
the C ABI of liborip.so (include/orip.h).  All compute runs in hand-written HIP kernels on gfx950; there is
no CPU fallback: importing `orip.lib` without the built library, or creating a Device without a GPU, raises.
"""
from .config import Config, load_config  # noqa: F401
