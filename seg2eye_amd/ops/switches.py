"""Experiment switches of the host side (environment, read once at import; the defaults are the shipped configuration).  They exist
for same-box A/B runs -- the pool's boxes differ by +-3 %, more than most single changes -- and tests flip the attributes of THIS
module (`ops.switches.SPARSE_OFF = True`) to run both sides of a path in one process.

Retired in round 5, their question answered (DESIGN appendix): S2E_CONV_STATS (InstanceNorm statistics from the conv epilogue: on),
S2E_WGRAD_C8_BATCH (the mlp_shared weight gradients as one launch: on), S2E_SPADE_PREPASS (label convs / class tables batched: on),
S2E_FC_HEAD (the encoder's head kernel: on), S2E_SPARSE_BWD_MIN (96) and S2E_SPADE_SPARSE_RECTS (256) are constants now."""
import os

_DET = os.environ.get('S2E_DETERMINISTIC', '0') == '1'          # every gradient summed in a fixed order (csrc: s2e_deterministic)

# 0 = every patch-resident weight gradient as its own launch (round 4); S2E_DETERMINISTIC keeps the fixed-order per-layer path
WGRAD_BATCH_OFF = os.environ.get('S2E_WGRAD_BATCH', '1') == '0' or _DET
# 0 = every generic weight gradient (1x1, stride-2, 4x4, small maps) as its own launch + reduction (rounds 1-5)
WGRAD_MULTI_OFF = os.environ.get('S2E_WGRAD_MULTI', '1') == '0'
# 0 = the dense SPADE backward of round 3 (the label-sparse one sums with float atomics: off under S2E_DETERMINISTIC)
SPARSE_BWD_OFF = os.environ.get('S2E_SPADE_SPARSE_BWD', '1') == '0' or _DET
# 0 = the dense fused launch everywhere (no label-uniform rectangles served from the class table)
SPARSE_OFF = os.environ.get('S2E_SPADE_SPARSE', '1') == '0'
# 0 = the two-launch path ([gamma | beta] conv, then the modulation) everywhere
FUSED_OFF = os.environ.get('S2E_SPADE_FUSED', '1') == '0'
# 1 = a flush runs its two big weight-gradient launches on two streams (round 6 experiment: 16.35 -> 16.29 ms, but the persistent batched launch
# holds the CUs and the small reduction launch behind it then shows 0.7 ms in the kernel trace: off, the accounting stays readable)
FLUSH_STREAMS = os.environ.get('S2E_FLUSH_STREAMS', '0') == '1'
# 0 = mlp_shared's weight gradient of a label-sparse SPADE backward walks the whole map (and d(actv) is zero-filled outside the work rectangles)
C8_SPARSE_OFF = os.environ.get('S2E_C8_SPARSE', '1') == '0'
