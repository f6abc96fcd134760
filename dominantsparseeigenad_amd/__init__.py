"""MI355X-native dominant-eigenpair autograd primitives (drop-in for DominantSparseEigenAD).

Sub-modules mirror the reference package layout (reference DominantSparseEigenAD/__init__.py is
empty; users import sub-modules by name):

    dominantsparseeigenad_amd.Lanczos   Lanczos, symeigLanczos
    dominantsparseeigenad_amd.CG        CG_torch, CGSubspace, setCGSubspaceSparse
    dominantsparseeigenad_amd.symeig    DominantSymeig, setDominantSparseSymeig
    dominantsparseeigenad_amd.eig       DominantEig, setDominantSparseEig
    dominantsparseeigenad_amd.operators TFIMOperator, CSROperator, Stencil3Operator (native mat-vecs)

The top-level ``DominantSparseEigenAD`` package in this repository re-exports the same modules
under the reference's import names.
"""
__version__ = "0.1.0"
