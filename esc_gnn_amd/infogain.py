"""
Drop-in for the reference's infogain wavelet-scale selection (/root/reference/models/infogain.py):

    from esc_gnn_amd.infogain import calc_infogain_wavelet_scales

same name, arguments, return values and error behaviour as models/infogain.py:172-664, computed on the GPU:

  * the powers P^t x, t = 0..T (the Python loop of T torch.sparse.mm calls + stack, :397-404) are ONE launch of the
    scattering chain with the all-powers event table {step t -> slot t}: `diffusion_powers` returns them as
    (N, C, T+1), step fastest, and nothing is re-laid out afterwards;
  * the per-graph divergences (the loops over graphs x channels x steps of :406-490, :780-900 and the nn_utilities
    helpers they call) are one kernel, escat_infogain_divergence_f32 (csrc/escat_infogain.hip): one workgroup per
    graph, no per-graph masks or copies;
  * what is left (:497-664: weights, sum over graphs, cumulative sums, quantile crossings, clean-up of uninformative
    channels) works on a (T, C) table and stays in torch.

Batches are anything with the attributes the reference reads from a torch_geometric Batch: batch[feat_to_diffuse_key],
batch[diff_op_key] (a torch sparse tensor or an esc_gnn_amd.ScatterOperator), .y, .num_graphs and .batch (sorted graph
index per node, as PyG collates it) -- or .ptr / .graph_ptr offsets.  The 'ot' metrics need the POT library and a
shortest-path matrix per graph (:85-128, :867-884); they are not part of the accelerated path and raise.
There is no CPU path: the HIP library must be built and the tensors must live on the GPU.
"""
import ctypes
import warnings
from typing import Dict, Iterable, Optional, Tuple
import torch

from . import _lib
from .pyg_utilities import bank, _ptr, _stream
from .operators import as_operator
from .scales import ChainPlan

DIVERGENCE_TYPES_MAP = {
    "kl": ("kl", "kld", "kl_divergence"),
    "l1": ("l1", "l1_diff", "tv", "total_variation"),
    "ot": ("ot", "wasserstein", "emd", "earth_mover"),
}


def all_powers_plan(T: int) -> ChainPlan:
    return ChainPlan(T, T + 1, tuple((t, t, +1, -1) for t in range(T + 1)))


def diffusion_powers(x: torch.Tensor, P_sparse, T: int) -> torch.Tensor:
    """(N, C) -> (N, C, T+1): [x, Px, P^2 x, ..., P^T x] along the last axis (the list of models/infogain.py:397-404,
    step-fastest instead of stacked on dim 0)."""
    if not x.is_cuda:
        raise RuntimeError("escat: x must live on a GPU; the MI355X path has no CPU fallback")
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError("escat: x must be a 2-D float32 tensor")
    op = as_operator(P_sparse)
    if op.n != x.shape[0]:
        raise RuntimeError(f"escat: size mismatch, operator is {op.n}x{op.n} but x has {x.shape[0]} rows")
    with torch.no_grad():
        return bank(x, op, all_powers_plan(int(T)))


def divergence_by_graph(Ptxs: torch.Tensor, graph_ptr: torch.Tensor, Ptx_i_start: int = 2, divergence_metric: str = "kl",
                        above_zero_floor: Optional[float] = None, check: bool = True,
                        max_graph_rows: Optional[int] = None) -> torch.Tensor:
    """get_divergence_by_channel (models/infogain.py:810-900) for every graph of a batch at once.
    Ptxs (N, C, T+1) from diffusion_powers; graph_ptr (B+1) row offsets -> (B, T - Ptx_i_start, C).
    max_graph_rows: the largest graph's rows if already known (saves one device->host read); 0 forces the
    three-pass kernel."""
    metric = divergence_metric.lower()
    if metric in DIVERGENCE_TYPES_MAP["ot"]:
        raise NotImplementedError("escat: optimal-transport divergences (POT library, per-graph shortest-path cost "
                                  "matrices) are outside the accelerated path")
    if metric not in DIVERGENCE_TYPES_MAP["kl"] + DIVERGENCE_TYPES_MAP["l1"]:
        raise ValueError(f"Unsupported divergence_metric '{divergence_metric}'. Choose 'kl', 'l1', or 'ot'.")
    lib = _lib.load()
    N, C, F = Ptxs.shape
    B = graph_ptr.numel() - 1
    Wc = Ptxs.contiguous()
    gp = graph_ptr.to(device=Ptxs.device, dtype=torch.int32).contiguous()
    out = torch.empty(B, F - 1 - Ptx_i_start, C, dtype=torch.float32, device=Ptxs.device)
    status = torch.empty(max(B, 1), dtype=torch.int32, device=Ptxs.device)
    if max_graph_rows is None:
        max_graph_rows = int((gp[1:] - gp[:-1]).max()) if B else 0
    if B and (int(gp[-1]) > N or max_graph_rows < 0):
        raise ValueError("escat: graph_ptr runs past the rows of Ptxs")
    with torch.cuda.device(Ptxs.device):
        rc = lib.escat_infogain_divergence_f32(_ptr(Wc), _ptr(gp), B, C, F, int(Ptx_i_start),
                                               0 if metric in DIVERGENCE_TYPES_MAP["kl"] else 1,
                                               0 if above_zero_floor is None else 1,
                                               ctypes.c_float(0.0 if above_zero_floor is None else float(above_zero_floor)),
                                               int(max_graph_rows), _ptr(out), _ptr(status), _stream(Ptxs.device))
    _lib.check(rc, "escat_infogain_divergence_f32")
    if check and B and bool(status[:B].any()):
        bad = torch.nonzero(status[:B]).reshape(-1).tolist()
        raise TypeError(f"infogain: a diffusion step of a channel in graph(s) {bad[:8]} cannot be normalised to a "
                        "probability vector (no variance, or a zero floor); the reference fails here too "
                        "(norm_tensor_to_prob_mass returns None, models/infogain.py:796)")
    return out


def _graph_ptr_of(batch, n_rows: int, num_graphs: int, device) -> torch.Tensor:
    for name in ("graph_ptr", "ptr"):
        p = getattr(batch, name, None)
        if p is not None:
            return torch.as_tensor(p).to(device=device, dtype=torch.int64)
    if num_graphs > 1:
        bi = batch.batch.to(device)
        if bi.numel() > 1 and bool((bi[1:] < bi[:-1]).any()):
            raise ValueError("escat: batch.batch must be sorted by graph (as torch_geometric collates it)")
        cnt = torch.bincount(bi, minlength=num_graphs)
        return torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)])
    return torch.tensor([0, n_rows], dtype=torch.int64, device=device)


def get_avg_infogain_wavelet_scales(channels_t_is: torch.Tensor, average_method: str = 'median') -> torch.Tensor:
    """models/infogain.py:726-757"""
    if average_method == 'median':
        return torch.median(channels_t_is, dim=0).values.to(torch.long)
    raise NotImplementedError(f"Averaging method '{average_method}' not implemented.")


def process_uninformative_channels(channels_t_is: torch.Tensor, strategy: str = "drop", T: int = 32) -> Tuple[torch.Tensor, torch.Tensor]:
    """models/infogain.py:667-723 (rows holding -1 are uninformative channels)."""
    uninform = torch.unique(torch.argwhere(channels_t_is == -1)[:, 0])
    if strategy == 'drop':
        keep = torch.ones(len(channels_t_is), dtype=torch.bool, device=channels_t_is.device)
        keep[uninform] = False
        channels_t_is = channels_t_is[keep]
        print(f"\tDropped {len(uninform)} uninformative channels: {uninform}")
    elif strategy in ('average', 'avg'):
        channels_t_is[uninform] = get_avg_infogain_wavelet_scales(channels_t_is)
    elif strategy == 'dyadic':
        n_ts = channels_t_is.shape[1]
        dyadic = torch.cat((torch.tensor([0]), 2 ** torch.arange({8: 3, 16: 4, 32: 5, 64: 6}[T] + 1)))
        if n_ts > len(dyadic):      # left-padded with (float) zeros, like the reference: needs a float table
            dyadic = torch.cat((torch.zeros(n_ts - len(dyadic)), dyadic))
            channels_t_is[uninform] = dyadic.to(channels_t_is.device)
    return uninform, channels_t_is


def select_scales(divs_by_graph: torch.Tensor, targets: Optional[torch.Tensor], task: str, T: int, n_channels: int,
                  device, cmltv_divergence_quantiles=(0.2, 0.4, 0.6, 0.8), start_from_t2: bool = True, failure_value=-1,
                  reweight_divs_for_target_imbal: bool = True, vector_feat_dim: Optional[int] = None) -> torch.Tensor:
    """models/infogain.py:497-653 on the (n_graphs, T - start, C) table of per-graph divergences."""
    start = 2 if start_from_t2 else 0
    divs = divs_by_graph
    if reweight_divs_for_target_imbal:
        if 'bin' in task.lower() and 'class' in task.lower():
            t = targets.to(divs.device)
            ones = t.sum()
            w = torch.ones(len(t), device=divs.device)
            w[t == 1] = ((t.shape[0] - ones) / ones).to(w.dtype)
            divs = torch.einsum('bTc,b->bTc', divs, w)
        else:
            warnings.warn(f"Reweighting KLDs not implemented for task='{task}'.")
    cum = torch.cumsum(divs.sum(dim=0), dim=0).cpu()                    # (T - start, C): tiny, finish on the host
    for c in range(n_channels):
        col = cum[:, c]
        rng = col[-1] - col[0]
        cum[:, c] = (col - col[0]) / rng if abs(float(rng)) >= 1e-10 else -torch.ones(cum.shape[0])
    rows = []
    for c in range(n_channels):
        r = []
        for q in cmltv_divergence_quantiles:
            hit = torch.nonzero(cum[:, c] >= q)
            r.append(int(hit[0, 0]) + start if hit.numel() else failure_value)
        rows.append(r)
    ts = torch.tensor(rows, device=device)
    if start_from_t2:
        prefix = torch.arange(3, device=device).unsqueeze(0).repeat(n_channels, 1)
        suffix = torch.full((n_channels, 1), T, device=device)
        ts = torch.cat((prefix, ts, suffix), dim=1).long()
    return ts


def calc_infogain_wavelet_scales(
    pyg_graphs,
    task: str,
    device: str,
    divergence_metric: str = "kl",
    divergence_metric_kwargs: Dict = {},
    diff_op_key: str = 'P',
    feat_to_diffuse_key: str = 'x',
    vector_feat_dim: Optional[int] = None,
    num_cuda_streams: int = 0,
    T: int = 16,
    cmltv_divergence_quantiles: Iterable[float] = (0.2, 0.4, 0.6, 0.8),
    start_from_t2: bool = True,
    failure_value=-1,
    reweight_divs_for_target_imbal: bool = True,
    data_subsample_prop: Optional[float] = None,
    data_subsample_random_state=None,
    auto_process_uninformative_channels: bool = True,
    uninformative_channel_strategy: str = 'drop',
    savepath_divergence_by_channel_plot: Optional[str] = None,
    plot_dyadic_kld_curve: bool = False,
    divergence_by_channel_plot_name: str = "infogain_wavelet_scales_plot",
    divergence_by_channel_plot_label: str = "",
    verbosity: int = 0,
):
    """See models/infogain.py:172-316 for the arguments.  Returns (uninformative channel indices | None, scales).
    `num_cuda_streams` is accepted and ignored (every graph of a batch is one workgroup of the same launch); the
    plotting arguments are accepted and ignored (matplotlib output is outside the hot path)."""
    start = 2 if start_from_t2 else 0
    if isinstance(pyg_graphs, dict):
        graphs, multiple = pyg_graphs['train'], True
    elif hasattr(pyg_graphs, 'to') and not isinstance(pyg_graphs, (list, tuple)):
        graphs, multiple = (pyg_graphs,), False                      # one Data object (single-graph task, :326-329)
    else:
        graphs, multiple = pyg_graphs, True                          # a loader: any iterable of batches
    floor = divergence_metric_kwargs.get("above_zero_floor") if divergence_metric_kwargs else None

    divs, targets, n_channels = [], [], None
    for batch_i, batch in enumerate(graphs):
        batch = batch.to(device)
        x = batch[feat_to_diffuse_key]
        d = 1
        if vector_feat_dim is not None and x.ndim == 2 and x.shape[1] == vector_feat_dim:
            x, d = x.reshape(-1, 1), vector_feat_dim
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        if getattr(x, 'is_sparse', False):
            x = x.to_dense()
        print(f"\tbatch {batch_i + 1} (shape={x.shape})")
        n_channels = x.shape[1]
        if multiple:
            num_graphs = batch.num_graphs
            targets.extend(batch.y)
        else:
            num_graphs = 1
            targets.extend(batch.y[batch.train_mask])
        if diff_op_key is None:
            # the reference's call here passes an argument get_P does not take (models/infogain.py:383-387 vs
            # data_processing/process_pyg_data.py:499-504) and cannot run; build the lazy walk it documents instead
            from .process_pyg_data import _get_P_csr
            from .operators import ScatterOperator
            n_nodes = x.shape[0] // d
            rp, col, val, _ = _get_P_csr(batch.edge_index, n_nodes, getattr(batch, 'edge_weight', None), True, x.device)
            P_sparse = ScatterOperator.from_csr(rp, col, val, n_nodes)
            if d != 1:
                raise ValueError("escat: a vector feature needs its Q operator (diff_op_key)")
        else:
            P_sparse = batch[diff_op_key]
        gp = _graph_ptr_of(batch, x.shape[0] // d, num_graphs, x.device) * d
        if num_graphs > 1 and data_subsample_prop is not None:
            import numpy as np
            rs = data_subsample_random_state if data_subsample_random_state is not None else np.random.RandomState(None)
            n_take = int(data_subsample_prop * num_graphs)
            if n_take > num_graphs + 1:
                raise ValueError(f"Cannot sample {n_take} unique integers from the range 0 to {num_graphs}.")
            pick = rs.choice(np.arange(0, num_graphs + 1), size=n_take, replace=False).tolist()   # utilities.py:46-86
            if max(pick, default=0) >= num_graphs:
                raise IndexError(f"sampled graph index {num_graphs} is past the batch (the reference samples from "
                                 f"0..num_graphs inclusive, models/infogain.py:369-373)")
        else:
            pick = None
        Ptxs = diffusion_powers(x.float().contiguous(), P_sparse, T)                       # (rows, C, T+1)
        dv = divergence_by_graph(Ptxs, gp, start, divergence_metric, floor)                # (B, T - start, C)
        divs.append(dv if pick is None else dv[torch.as_tensor(pick, device=dv.device)])
    divs = torch.cat(divs, dim=0)
    targets = torch.stack([torch.as_tensor(t) for t in targets]) if len(targets) else None
    if verbosity > 0:
        print('klds_by_x_t_chan.shape:', tuple(divs.shape))

    channels_t_is = select_scales(divs, targets, task, T, n_channels, device, cmltv_divergence_quantiles, start_from_t2,
                                  failure_value, reweight_divs_for_target_imbal, vector_feat_dim)
    if verbosity > 0:
        print('channels_t_is.shape:', channels_t_is.shape)
        print('channels_t_is\n', channels_t_is)
    failed = torch.unique(torch.argwhere(channels_t_is == failure_value)[:, 0])
    if (not auto_process_uninformative_channels) and len(failed) > 0:
        warnings.warn(f"The channels/features at indexes {[i.item() for i in failed]} failed to generate valid "
                      f"diffusion scales. Consider dropping these features?")
    if vector_feat_dim is None and bool((channels_t_is[0] == channels_t_is).all()):
        warnings.warn("All channels' custom scales were found to be equal: returning 1-d tensor of scales instead.")
        channels_t_is = channels_t_is[0]
    if auto_process_uninformative_channels:
        return process_uninformative_channels(channels_t_is, uninformative_channel_strategy, T)
    return None, channels_t_is
