"""midas_amd -- MI355X-native pileup + allele counting for `run_midas.py snps`.

Only the hot path of snayfach/MIDAS named in BASELINE.json lives here:
``midas/run/snps.py`` pileup (index_bam -> pysam_pileup -> snps_summary), as
hand-written HIP kernels for gfx950 behind a C-ABI (include/midas_snps.h), with a
Python host that mirrors the reference's own interface for that path.
"""

__version__ = "0.1.0"
