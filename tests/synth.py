"""re-export of the package's synthetic-input generators for the tests"""
import importlib

_s = importlib.import_module("watermarking-gpu_amd.synth")
synth_frame = _s.synth_frame
synth_watermark = _s.synth_watermark
SEED = _s.SEED
