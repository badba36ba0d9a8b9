"""The exchange step on the GPU backend: torch.distributed "nccl" is RCCL on ROCm.  One GPU per box here, so the collective
runs at world_size 1 -- it still goes through RCCL's communicator set-up and all_gather_into_tensor on device buffers (the
multi-rank ordering / ragged logic is covered by the gloo world-2 tests in test_host_cpu.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_packed_gather_under_rccl_world1():
    import torch.distributed as dist
    from ullsam_amd import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        low = torch.randn(4, 1, 256, 256, device="cuda")
        mk = (low[:, :, :64, :64] > 0).to(torch.uint8).repeat(1, 1, 16, 16).contiguous()
        tok = torch.arange(4 * 6, device="cuda").reshape(4, 6)
        a, b, c = parallel.gather_mask_results(low, mk, tok, counts=[4])
        assert torch.equal(a, low) and torch.equal(b, mk) and torch.equal(c, tok)
        pend = parallel.gather_mask_results_async(low, mk, None)          # counts exchanged by a collective of its own
        a2, b2, c2 = pend.wait()
        torch.cuda.synchronize()
        assert torch.equal(a2, low) and torch.equal(b2, mk) and c2 is None
        assert parallel.world() == (0, 1)
    finally:
        dist.destroy_process_group()
