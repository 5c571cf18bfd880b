"""phyloformer_amd — MI355X-native Phyloformer inference (one-hot MSA → pairwise distances).

Host code (this package) is Python; all arithmetic lives in
``libphyloformer_amd.so`` (hand-written HIP for gfx950) behind the C ABI in
``include/phyloformer_amd.h``.  See DESIGN.md.
"""
from .ckpt import CheckpointError, load_ckpt, load_state_dict  # noqa: F401
from .fasta import ALPHABET, load_alignment, one_hot, parse_fasta  # noqa: F401
from .phylip import vec_to_phylip  # noqa: F401
from .weights import ModelWeights, from_state_dict, load_weights  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # the engine pulls in ctypes + the native library: import it lazily so the
    # pure-host helpers above work on machines without the .so
    if name in ("Engine", "EngineError", "load_library"):
        from . import engine
        return getattr(engine, name)
    if name == "Phyloformer":
        from .model import Phyloformer
        return Phyloformer
    raise AttributeError(name)
