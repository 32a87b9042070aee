"""GPU: neighbour pooling / vector invariants against the CPU restatement of models/escgnn_modular.py:403-504."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(N, E, seed, isolated=()):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, N, (E,), generator=g); dst = torch.randint(0, N, (E,), generator=g)
    keep = torch.ones(E, dtype=torch.bool)
    for n in isolated:
        keep &= dst != n
    return torch.stack([src[keep], dst[keep]])


@pytest.mark.parametrize("stats,stride", [(('percentiles', 'min', 'max', 'mean'), 0.2), (('mean', 'var'), 0.2),
                                          (('quantiles', 'var', 'max'), 0.25), (('min',), 0.2), ((), 0.2),
                                          (('percentiles',), 0.34)])
@pytest.mark.parametrize("d", [3, 2, 4])          # 2, 3: fused kernel; 4: torch cosines + pooling kernel
def test_vector_invariants_forward_and_gradient(stats, stride, d):
    from esc_gnn_amd.invariants import get_vector_invariants
    from oracle.invariants_oracle import ref_vector_invariants
    dev = torch.device("cuda:0")
    N, Wp = 300, 7
    ei = _graph(N, 2400, seed=len(stats) + int(stride * 100), isolated=(0, 17, 299))
    g = torch.Generator().manual_seed(1)
    Wv = torch.randn(N, Wp, 1, d, generator=g)
    Wz = Wv.clone(); Wz[5] = 0.0                                          # a zero vector: clamped norm in the cosine
    ref0 = ref_vector_invariants(Wz, ei, stats, stride)
    out0 = get_vector_invariants(Wz.to(dev), ei.to(dev), stats, stride)
    assert out0.shape == ref0.shape and (out0.cpu() - ref0).abs().max() <= 2e-6
    ref_in = Wv.clone().requires_grad_(True)
    ref = ref_vector_invariants(ref_in, ei, stats, stride)
    x = Wv.to(dev).requires_grad_(True)
    out = get_vector_invariants(x, ei.to(dev), stats, stride)
    assert out.shape == ref.shape
    assert (out.cpu() - ref).abs().max() <= 2e-6, float((out.cpu() - ref).abs().max())
    G = torch.randn(ref.shape, generator=g)
    ref.backward(G); out.backward(G.to(dev))
    err = (x.grad.cpu() - ref_in.grad).abs().max()
    assert err <= 2e-5 * max(1.0, float(ref_in.grad.abs().max())), float(err)


def test_pool_kernel_ties_empty_and_limits():
    """equal values (first in edge order wins), empty segments give 0, degree limit is an error code."""
    from esc_gnn_amd.invariants import _NeighborPool, _k_table, quantile_levels
    from esc_gnn_amd import EscatError
    dev = torch.device("cuda:0")
    vals = torch.tensor([[1.0], [1.0], [0.5], [2.0], [2.0]], device=dev, requires_grad=True)
    seg = torch.tensor([0, 3, 3, 5], dtype=torch.int32, device=dev)      # node 0: 3 edges, node 1: none, node 2: 2 edges
    kt = _k_table(quantile_levels(0.5)).to(dev)                           # one level, p = 0.5
    out = _NeighborPool.apply(vals, seg, 3, 1 | 2 | 4 | 8, kt, 3)
    want = torch.tensor([[2.5 / 3, 1.0, 0.5, (2.25 / 3) - (2.5 / 3) ** 2, 1.0], [0, 0, 0, 0, 0], [2.0, 2.0, 2.0, 0.0, 2.0]])
    assert torch.allclose(out.cpu(), want, atol=1e-6)
    out.backward(torch.tensor([[0.0, 1.0, 0.0, 0.0, 10.0], [1.0] * 5, [0.0, 1.0, 100.0, 0.0, 0.0]], device=dev))
    # node 0: max -> first 1.0 (edge 0); median k = int(2*0.5) = 1 -> sorted [0.5, 1(e0), 1(e1)][1] = edge 0
    assert vals.grad.flatten().tolist() == [11.0, 0.0, 0.0, 101.0, 0.0]
    with pytest.raises(EscatError):
        _NeighborPool.apply(vals.detach(), seg, 3, 1, kt, 65)
