"""Reproducer of the round-4 GPU-suite abort.  ProcessGroupNCCL's watchdog thread polls the end event of every
EAGER collective with hipEventQuery (every 100 ms, until it has seen the event complete).  This script issues an eager all-reduce and
then, inside that polling period, opens a stream capture that CONTAINS a collective of a process group -- the pattern of
engine.TrainStep after its eager warm-up steps.

    python tools/repro_watchdog_capture.py VARIANT [rounds]
      plain-global        capture without a collective, global error mode
      coll-global         captured collective on the SAME group as the eager one, torch's default (global) mode
      coll-thread_local   same, capture_error_mode="thread_local"
      coll-relaxed        same, capture_error_mode="relaxed"
      coll-drain          same as coll-global after a 0.3 s pause (the watchdog has retired the eager work)
      coll-owngroup       the captured collective runs on a group of its own (dist.new_group): the eager group's stream never captures
      sameside-global     the eager all-reduce is issued ON the stream that is captured next (torch runs a synchronous collective on the
                          current stream and records its end event there) -- GradBucket's side stream in round 4
      sameside-thread_local / sameside-relaxed / sameside-drain      same, other error modes / after the pause
      othercoll-MODE      otherside with a collective inside the capture as well (the shape of the fixed GradBucket)
      otherside-global    the eager all-reduce on a second side stream that never captures (the round-5 fix)

One rank, one GPU; none of this package's kernels is involved."""
import os
import sys
import time

import torch
import torch.distributed as dist

variant = sys.argv[1] if len(sys.argv) > 1 else "coll-global"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()
side2 = torch.cuda.Stream()
kind, _, mode = variant.partition("-")
cap_group = dist.new_group([0], backend="nccl") if mode == "owngroup" else None
if cap_group is not None:
    dist.all_reduce(y, group=cap_group)                 # communicator created outside any capture
    torch.cuda.synchronize()
    time.sleep(0.3)
error_mode = mode if mode in ("global", "thread_local", "relaxed") else "global"
for r in range(rounds):
    if kind in ("sameside", "otherside", "othercoll"):
        es = side if kind == "sameside" else side2
        es.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(es):
            dist.all_reduce(x)
        torch.cuda.current_stream().wait_stream(es)
    else:
        dist.all_reduce(x)                              # eager: a Work item the watchdog polls until it has seen it complete
    torch.cuda.synchronize()                            # the GPU is idle -- the watchdog has not polled yet
    if mode == "drain":
        time.sleep(0.3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode=error_mode):
        y.add_(1.0)
        if kind in ("coll", "othercoll"):
            dist.all_reduce(y, group=cap_group)
        time.sleep(0.25)                                # the capture stays open across at least two watchdog polls
        y.mul_(0.5)
    g.replay()
    torch.cuda.synchronize()
    print(f"round {r}: {variant} survived", flush=True)
dist.destroy_process_group()
print("OK")
