"""Child process of tests/test_closures_gpu.py::test_rccl_two_ranks: one rank of a world of N on its own GPU (started fresh, before anything
touched the card).  Decodes its contiguous shard of the users with BSSD_batch on the golden case's models, all-gathers the counters over RCCL
(backend "nccl"), writes what it produced to <out>.<rank>.json.   usage: python -m tests.rccl_worker <rank> <world> <port> <n_users> <out>"""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world, port, n_users, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from atspeed_amd import synth
    from atspeed_amd.beamSD import BSSD_batch
    from atspeed_amd.dist import Counters, all_gather_counters, shard_range
    from atspeed_amd.model import HipLlama
    from tests.golden.cases import CASES, build_case_inputs
    dev = torch.device("cuda", rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    case = [c for c in CASES if c["name"] == "k5_dk10_indep"][0]
    ci = build_case_inputs(case)
    kw = dict(max_slots=512, max_tokens=512, max_logit_rows=448, device=dev)
    tgt = HipLlama.from_state_dict(ci["target_dims"], ci["target_sd"], torch.float32, num_beams=case["K"], **kw)
    drf = HipLlama.from_state_dict(ci["draft_dims"], ci["draft_sd"], torch.float32, num_beams=case["DK"], **kw)
    lo, hi = shard_range(n_users, rank, world)
    prompts = [synth.synthetic_prompt(16 + u, 100 + u) for u in range(lo, hi)]
    ins = [{"input_ids": torch.from_numpy(p)[None].to(dev)} for p in prompts]
    res = BSSD_batch(tgt, drf, ins, case["gamma"], case["max_new_tokens"], prefix_allowed_tokens_fn=ci["fn"]) if ins else []
    c = Counters()
    for o in res:
        c.add_result(o)
    c.elapsed_ns = 1_000_000 * (rank + 1)
    allc = all_gather_counters(c, dev)
    P = [len(p) for p in prompts]
    with open(f"{out}.{rank}.json", "w") as f:
        json.dump(dict(users=list(range(lo, hi)), tokens=[o["beam_sequence"][:, p:].cpu().tolist() for o, p in zip(res, P)],
                       n_run=[o["n_run"] for o in res], accept=[o["accept_steps"] for o in res],
                       gathered=[(x.n_users, x.n_run, x.accept_steps, x.elapsed_ns) for x in allc]), f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
