"""consent_amd -- MI355X-native window-correction engine for CONSENT (host-side mirror of the operator seam).

The compute lives in ``libconsent_amd.so`` (HIP, gfx950), reached through the C ABI in ``include/consent_amd.h``.
This package only marshals buffers; it never computes a consensus itself and raises if the library is missing.
"""
from .engine import (  # noqa: F401
    Engine,
    EngineError,
    PafReader,
    Params,
    ReadIndex,
    SynthSpec,
    WIN_CONSENSUS,
    WIN_OVERFLOW,
    WIN_TEMPLATE,
    compute_consensus_read_correction,
    compute_consensus_assembly_polishing,
    lib_path,
    load_library,
    pack_piles,
    paf_explode,
    paf_merge,
    paf_reformat,
)
