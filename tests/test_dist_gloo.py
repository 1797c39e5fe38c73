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
        info = cdist.broadcast_weights_ex(model, src=0)
        assert info["transport"] == "host" and info["nranks"] is None and info["broadcast_ms"] > 0.0   # CPU ranks: the host blob over gloo
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


def _negotiation_worker(rank, world, port, out_dir, scenario):
    """Drive WeightsComm.negotiate (the transport choice of broadcast_weights on GPU ranks) over gloo with the two C-ABI
    steps replaced: every scenario must end with the SAME answer on both ranks and with no rank left inside a collective."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from celebrity_image_denoiser_amd import dist as cdist

        class FakeLib:
            destroyed = []

            def cid_comm_destroy(self, comm):
                FakeLib.destroyed.append(comm)
                return 0

        cdist._lib.lib = lambda: FakeLib()                    # negotiate() only needs cid_comm_destroy from it here
        cdist._rccl_available = lambda: not (scenario == "rccl_missing_on_rank1" and rank == 1)

        def make_id():
            if scenario == "unique_id_fails_on_rank0":
                raise RuntimeError("ncclGetUniqueId refused")
            return bytes(range(128))

        def init_rank(device, world_, ident, rank_):
            assert ident == bytes(range(128)) and world_ == world and rank_ == rank
            if scenario == "init_fails_on_rank1" and rank == 1:
                raise RuntimeError("ncclCommInitRank refused")
            return f"comm{rank}"

        cdist.WeightsComm._make_unique_id = staticmethod(make_id)
        cdist.WeightsComm._init_rank = staticmethod(init_rank)
        if scenario == "cached":
            # the communicator is made once per (group, device) and kept: the second request makes no collective
            comm, fresh = cdist.WeightsComm.for_group(torch.device("cpu"))
            again, fresh2 = cdist.WeightsComm.for_group(torch.device("cpu"))
            assert fresh and not fresh2 and again is comm and comm is not None
        else:
            comm = cdist.WeightsComm.negotiate(torch.device("cpu"))
        got = "comm" if comm is not None else "none"
        # a collective AFTER the negotiation: it only completes if both ranks left negotiate() in step
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        closed = list(FakeLib.destroyed)
        if comm is not None:
            comm.close()
        open(os.path.join(out_dir, f"{scenario}_rank{rank}.txt"), "w").write(f"{got} sum={t.item():.0f} closed_early={closed}")
    finally:
        dist.destroy_process_group()


def test_transport_negotiation_never_leaves_a_rank_behind(tmp_path):
    """VERDICT r2 #8 / ADVICE r2: when RCCL (or the communicator) is unavailable on ONE rank, both ranks must fall back to
    transport 2 together.  The old code raised on rank 0 before the id exchange and left rank 1 inside broadcast_object_list."""
    want = {
        "all_fine": ("comm sum=3 closed_early=[]", "comm sum=3 closed_early=[]"),
        "cached": ("comm sum=3 closed_early=[]", "comm sum=3 closed_early=[]"),
        "rccl_missing_on_rank1": ("none sum=3 closed_early=[]", "none sum=3 closed_early=[]"),
        "unique_id_fails_on_rank0": ("none sum=3 closed_early=[]", "none sum=3 closed_early=[]"),
        # rank 0 did set its communicator up: it is destroyed again before falling back
        "init_fails_on_rank1": ("none sum=3 closed_early=['comm0']", "none sum=3 closed_early=[]"),
    }
    for scenario, expect in want.items():
        mp.spawn(_negotiation_worker, args=(2, _free_port(), str(tmp_path), scenario), nprocs=2, join=True)
        for r in range(2):
            assert open(os.path.join(tmp_path, f"{scenario}_rank{r}.txt")).read() == expect[r], (scenario, r)


def _subgroup_worker(rank, world, port, out_dir):
    """broadcast_weights inside a sub-group whose source is not global rank 0: `src` is a global rank (as in
    torch.distributed.broadcast) and must be translated to the group rank consistently."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import celebrity_image_denoiser_amd as cid
        from celebrity_image_denoiser_amd import dist as cdist, synth

        sub = dist.new_group([1, 2])
        ok = True
        if rank in (1, 2):
            torch.manual_seed(99 + rank)
            model = cid.DenoiseGenerator()
            if rank == 2:                                   # global rank 2 = group rank 1 owns the checkpoint
                model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("default").items()})
            cdist.broadcast_weights(model, src=2, group=sub)
            ref = synth.make_state_dict("default")
            ok = all(np.array_equal(model.state_dict()[k].numpy(), ref[k]) for k in ref)
        open(os.path.join(out_dir, f"sub_rank{rank}.txt"), "w").write("ok" if ok else "FAIL")
    finally:
        dist.destroy_process_group()


def test_broadcast_in_a_subgroup_from_a_nonzero_source(tmp_path):
    mp.spawn(_subgroup_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert open(os.path.join(tmp_path, f"sub_rank{r}.txt")).read() == "ok"
