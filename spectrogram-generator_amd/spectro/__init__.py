"""spectro -- MI355X-native STFT/PSD engine behind the reference's Python call surface.

    from spectro import spectrogram          # drop-in for scipy.signal.spectrogram (PlotEngine.py:8)

Device work goes through libspectro.so (include/spectro.h) via ctypes; see DESIGN.md.
"""
from .signal import spectrogram  # noqa: F401
from .windows import get_window  # noqa: F401

__version__ = "0.1.0"
