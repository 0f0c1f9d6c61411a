"""Import alias: the package directory is ``falcon-r1cs_amd/`` (not a valid Python identifier),
so ``import falcon_r1cs_amd`` resolves its submodules from there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "falcon-r1cs_amd"))

from ._lib import FrwError, lib_path, load_library  # noqa: E402,F401
from .engine import (G_ADD_MOD, G_L2_ELEM, G_LESS_THAN_Q, G_MOD_Q, G_NORM_BOUND_512,  # noqa: E402,F401
                     G_NORM_BOUND_1024, ENC_CANONICAL, ENC_COMPACT, ENC_MONTGOMERY, ST_COEFF_RANGE, ST_DECODE, ST_NORM_BOUND, ST_OK, NONCE_LEN, PK_LEN, SIG_LEN,  # noqa: E402,F401
                     CompactLayout, Groth16Verifier, Layout, MI355X_BENCH_LAUNCH_SHAPES, VERIFY_POINTS_ARE_CHECKED, VK_POINTS_ARE_CHECKED,
                     KEY_AUTO, KEY_TABLES, KEY_BARE, GROTH16_PARTIAL_WORDS, GROTH16_COMBINE_WORKSPACE, WitnessEngine, diag_pairing, compact_layout, layout, layout_dual, synth_triples)
