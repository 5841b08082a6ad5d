"""stark_amd -- MI355X (gfx950) implementation of the STARK prover hot path
(NTT, coset LDE, FRI split-and-fold, BLAKE2b Merkle commitment) over
p = 407*2^119 + 1, behind the C ABI declared in include/stark_hip.h."""
from .device import Context, FriLayers, StarkError, Tree  # noqa: F401

P = 407 * (1 << 119) + 1
GENERATOR = 85408008396924667383611388730472331217
