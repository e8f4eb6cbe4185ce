"""rec_now_amd -- MI355X (gfx950) native implementation of rec_now's in-batch ranking-loss and feature-interaction
hot path.  Module paths mirror the reference package (`rec_now.layers.*`, `rec_now.rec_block.*`, `rec_now.util.*`);
compute runs in hand-written HIP kernels behind the C ABI of include/recnow.h (librecnow_hip.so)."""
__version__ = '0.1.0'
