"""Import alias: the package directory is ``falcon-r1cs_amd/`` (not a valid Python identifier),
so ``import falcon_r1cs_amd`` resolves its submodules from there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "falcon-r1cs_amd"))

from ._lib import FrwError, lib_path, load_library  # noqa: E402,F401
from .engine import (ENC_CANONICAL, ENC_MONTGOMERY, ST_COEFF_RANGE, ST_NORM_BOUND, ST_OK,  # noqa: E402,F401
                     Layout, WitnessEngine, layout, synth_triples)
