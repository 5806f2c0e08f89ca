"""mitoflex_amd -- MI355X-native k-mer bait read pre-filter for the MitoFlex pipeline.

Only what the hot path needs lives here:
  csrc/                      HIP kernels (gfx950) + the C ABI of libmitofilter_hip.so
  mitofilter.py              thin ctypes wrapper over include/mitofilter.h
  assemble/assemble_wrapper  host-side mirror of the reference call site
                             (assemble/assemble_wrapper.py:105-108, 162-200, 317-345)
  assemble/fastfilter        built drop-in CLI (contig filter + `bait` sub-command)
  utility/helper.py          shell_call / concat_command / direct_call mirror
                             (utility/helper.py:35-86)
"""
__version__ = "0.1.0"
