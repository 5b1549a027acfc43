from x264vfw_amd.synth import *  # noqa: F401,F403
from x264vfw_amd.synth import synth_frames, psnr  # noqa: F401
