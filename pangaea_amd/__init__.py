"""pangaea_amd -- MI355X (gfx950) implementation of Pangaea's barcode-aware k-mer feature + VAE-encode path.

Host side stays Python and mirrors the reference's interfaces (``feature.Feature``, ``data.Data``,
``models.VAENET``, ``clustering.clustering_rph_kmeans``); the counting kernels are hand-written HIP behind the C
ABI of ``include/pangaea_feat.h`` (``libpangaea_feat.so``, loaded with ctypes).
"""
__version__ = "0.1.0"
