"""The N>1 path on CPU: world_size 2 over gloo.  Rank 0 owns the checkpoint; ONE broadcast of the
packed weights blob must leave rank 1 with identical parameters, and the contiguous shards of a batch
must tile it.  (On GPUs the same code runs over backend "nccl" = RCCL; the forward has no collective.)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import celebrity_image_denoiser_amd as cid
        from celebrity_image_denoiser_amd import dist as cdist, synth

        torch.manual_seed(1234 + rank)                      # different random init per rank
        model = cid.DenoiseGenerator()
        if rank == 0:
            model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("hot").items()})
        cdist.broadcast_weights(model, src=0)
        ref = synth.make_state_dict("hot")
        same = all(np.array_equal(model.state_dict()[k].numpy(), ref[k]) for k in ref)
        begin, end = cdist.shard_range(11, rank, world)
        x, _, _ = synth.make_batch(end - begin, 8, 8, first_index=begin)
        # every rank checksums its shard; the gathered list must equal the checksum of the whole batch
        mine = torch.tensor([float(x.astype(np.float64).sum()), float(end - begin)], dtype=torch.float64)
        parts = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine)
        whole, _, _ = synth.make_batch(11, 8, 8, first_index=0)
        ok = same and abs(sum(p[0].item() for p in parts) - float(whole.astype(np.float64).sum())) < 1e-6 \
            and sum(int(p[1].item()) for p in parts) == 11
        open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else f"FAIL same={same}")
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(os.path.join(tmp_path, f"rank{r}.txt")).read() == "ok"
