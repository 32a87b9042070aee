"""
esc_gnn_amd -- MI355X (gfx950) implementation of ESc-GNN's diffusion-wavelet
scattering hot path behind the reference's own Python signatures.

    get_Batch_Wjxs   <- pyg_utilities.get_Batch_Wjxs          (pyg_utilities.py:283)
    scatter          <- ESCGNNModular._scatter                (models/escgnn_modular.py:309)
    ScatterOperator  canonical device CSR / BSR-3 form of P / Q
    OperatorBatch    a mini-batch drawn (by part ids, no copy) from a stored ScatterOperator

The compute lives in libescat.so (esc_gnn_amd/csrc, C ABI in include/escat.h).
"""
from ._lib import EscatError, load as load_library            # noqa: F401
from .operators import ScatterOperator, OperatorBatch, as_operator           # noqa: F401
from .scales import ChainPlan, plan_chain                     # noqa: F401
from .pyg_utilities import get_Batch_Wjxs                     # noqa: F401
from .scatter import scatter, escat_scatter                   # noqa: F401
