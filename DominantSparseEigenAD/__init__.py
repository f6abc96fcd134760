"""Drop-in import name.  ``import DominantSparseEigenAD.symeig as symeig`` (reference README.md:59-67,
examples/schrodinger1D.py:3) resolves to the MI355X-native modules of ``dominantsparseeigenad_amd``;
the sub-modules are the very same module objects, so the ``set...``-then-attribute protocol works."""
import importlib
import sys

for _name in ("Lanczos", "CG", "symeig", "eig"):
    _mod = importlib.import_module("dominantsparseeigenad_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
del _name, _mod
