"""
esc_gnn_amd -- MI355X (gfx950) implementation of ESc-GNN's diffusion-wavelet
scattering hot path behind the reference's own Python signatures.

    get_Batch_Wjxs   <- pyg_utilities.get_Batch_Wjxs          (pyg_utilities.py:283)
    scatter          <- ESCGNNModular._scatter                (models/escgnn_modular.py:309)
    get_vector_invariants         <- ESCGNNModular._get_vector_invariants (models/escgnn_modular.py:403)
    calc_infogain_wavelet_scales  <- models.infogain.calc_infogain_wavelet_scales (models/infogain.py:172)
    process_pyg_data / build_PQ_batch  <- data_processing.process_pyg_data (esc_gnn_amd.process_pyg_data)
    ScatterOperator  canonical device CSR / BSR-3 form of P / Q
    OperatorBatch    a mini-batch drawn (by part ids, no copy) from a stored ScatterOperator

The compute lives in libescat.so (esc_gnn_amd/csrc, C ABI in include/escat.h).
"""
from ._lib import EscatError, load as load_library            # noqa: F401
from .operators import ScatterOperator, OperatorBatch, FactoredVectorOperator, as_operator           # noqa: F401
from .scales import ChainPlan, plan_chain                     # noqa: F401
from .pyg_utilities import get_Batch_Wjxs                     # noqa: F401
from .scatter import scatter, scatter_both, escat_scatter, ScatteringCache     # noqa: F401
from .both import get_Batch_Wjxs_both, BothTracksOperator     # noqa: F401
from .store import load_operator_store, ScatteringStore       # noqa: F401
from .invariants import get_vector_invariants                 # noqa: F401
from .infogain import calc_infogain_wavelet_scales            # noqa: F401
