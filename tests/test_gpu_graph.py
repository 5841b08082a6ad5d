"""The resident entry points under HIP graph capture: once the tables and the scratch of a size exist, a
transform launches kernels and nothing else, and the entry point that allocates nothing (the Merkle root
through the static scratch) does so from its first call.  A captured sequence is replayed on new data and
checked against the restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_c as oc          # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from stark_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_n", [8, 12, 16, 18])
def test_transform_and_static_root_replayed_from_a_graph(ctx, log_n):
    import torch
    n = 1 << log_n
    first, second = oc.splitmix(n, 31 + log_n), oc.splitmix(n, 91 + log_n)
    d = torch.from_numpy(first.view(np.int64).copy()).cuda()
    root = torch.zeros(32, dtype=torch.uint8, device="cuda")
    # one call of each size outside the capture builds its twiddle tables and scratch
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.ntt_dev(d, n, False)
    ctx.ntt_dev(d, n, True)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.ntt_dev(d, n, False)                                  # forward in place
        ctx.merkle_root_fe_static_dev(d, n, root)                 # commitment of the transformed vector
        ctx.ntt_dev(d, n, True)                                   # and back
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for x in (first, second, first):
            d.copy_(torch.from_numpy(x.view(np.int64).copy()))
            root.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert bytes(root.cpu().numpy()) == oc.merkle_commit_fe(oc.ntt(x))
            assert np.array_equal(d.cpu().numpy().view(np.uint64), x)
    finally:
        del graph
