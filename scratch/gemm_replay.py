"""Replay of the main stream's >= 1 GFLOP trailing-update launches of ONE flag-schedule evaluation, one at a time, for rocprofv3
--pmc passes (VERDICT r5 #2b: one launch population for roofline.flops_per_launch / traffic / algorithmic_bytes).

A counter pass runs one kernel at a time, so the library cannot keep its flag edges under it and falls back to the event
schedule (urgent and rest separate): the launches a --pmc pass of bench.py sees are NOT the ones the timed steps run.  Here the
shapes of the timed schedule's launches (logged by the library, GPT_GEMM_LOG, during an unprofiled evaluation) are launched
through gpt_dev_gemm_nt on the context's CU-masked main stream with the operand geometry of the factorisation: one NP x NP
matrix, C = the trailing block at (u0, u0), A = B = the panel columns left of it.  Under a counter pass every launch is alone
on the chip anyway, which is exactly what the replay is.

usage: gemm_replay.py <shape log> [repeats]        (prints the shapes' flops / algorithmic bytes as JSON on the last line)"""
import json
import sys

import torch

sys.path.insert(0, '/root/repo')
from gptools_amd import _lib  # noqa: E402

rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
rows = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(r[4]), int(r[5]) if len(r) > 5 else 1) for r in rows]
# the LAST evaluation in the log: its first launch is the largest of a decreasing run
start = 0
for i in range(1, len(rows)):
    if rows[i][0] > rows[i - 1][0]:
        start = i
shapes = [r for r in rows[start:] if r[5] == 1]            # (those that ran the 64x64 kernel)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NP = max(r[0] for r in shapes) + 1024
lib = _lib.load()
ctx = _lib.Context(0)
ctx.set_option("lookahead", 1)
st = torch.cuda.ExternalStream(int(ctx.stream))
with torch.cuda.stream(st):
    M = torch.randn(NP, NP, dtype=torch.float64, device="cuda")
    for _ in range(reps):
        for (m, n, k, tri, fl, k64) in shapes:
            u0 = NP - m
            c0 = u0 - k - 128
            A = M.data_ptr() + 8 * (u0 * NP + c0)
            Cp = M.data_ptr() + 8 * (u0 * NP + u0)
            _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, -1.0, A, NP, A, NP, 1.0, Cp, NP, tri))
st.synchronize()
fl = [s[4] for s in shapes]
by = [16.0 * s[4] / (2.0 * s[2]) + 8.0 * s[2] * max(s[0], s[1]) for s in shapes]
print("REPLAY " + json.dumps({"launches_per_evaluation": len(shapes), "repeats": reps, "flops_per_launch": sum(fl) / len(fl),
                              "algorithmic_bytes_per_launch": sum(by) / len(by), "shapes": [list(s[:4]) for s in shapes]}))
