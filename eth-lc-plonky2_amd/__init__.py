"""MI355X-native Plonky2 prover backend for the Ethereum light-client circuit.

Host-side mirror of the reference's prover boundary (build / prove / verify,
eth-lc-plonky2/src/main.rs:226-233) over the C ABI in include/lcp2.h.  The
directory name follows the project ("eth-lc-plonky2_amd"); import it as
`eth_lc_plonky2_amd` (shim module at the repository root).
"""
from .binding import (CircuitData, ProofRejected, Context, Lcp2Error, Oracle, Params, load_library, standard_params,  # noqa: F401
                      MEM_DEVICE, MEM_HOST, KERNEL_FAMILIES, GOLDILOCKS_P, proof_to_bytes, proof_from_bytes, proof_layout)
from .build import build_native, build_host  # noqa: F401
from . import binding  # noqa: F401,E402
from . import circuit  # noqa: F401,E402
from . import poseidon_py  # noqa: F401,E402
from . import recursion_gates  # noqa: F401,E402
from . import u32_gates  # noqa: F401,E402
from . import batch  # noqa: F401,E402
from . import parallel  # noqa: F401,E402
from . import light_client  # noqa: F401,E402
