"""Summarise the counter passes of scratch/r06_gemm_stalls.sh for gemm_nt_kernel<64,64>: where the wave-cycles of the dominant kernel go.
Units (MI355X_MICROARCH.md, 'rocprofv3 PMC slots'): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-pipe; SQ_BUSY_CYCLES per SE; GRBM_GUI_ACTIVE = shader-clock cycles of the dispatch."""
import csv, sys, collections, re, glob, os
src = sys.argv[1]
def load(d):
    p = os.path.join(d, '**', 't_counter_collection.csv')
    f = glob.glob(p, recursive=True)
    if not f: return None
    rows = list(csv.DictReader(open(f[0])))
    acc = collections.defaultdict(float); disp = set(); dur = 0.0; seen = set()
    for r in rows:
        if 'gemm_nt_kernel' not in r['Kernel_Name'] or '64, 64' not in r['Kernel_Name'] or int(r["Grid_Size"]) < 54528:
            continue        # the >= 1 GFLOP launches of the 64x64-tile kernel (213 workgroups x 256 threads and up)
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); dur += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    acc['_launches'] = len(seen); acc['_dur_ns'] = dur
    return acc
for group in ('one_m7168', 'one_m4096', 'bench'):
    tot = {}
    for d in sorted(glob.glob(os.path.join(src, group + '_p*'))):
        if not os.path.isdir(d): continue
        a = load(d)
        if a is None: continue
        tag = d.rsplit('_', 1)[1]
        for k, v in a.items():
            tot[(tag, k)] = v
    if not tot: continue
    print("== %s: gemm_nt_kernel<64,64>, launches >= 1 GFLOP" % group)
    for tag in sorted(set(t for t, _ in tot)):
        n = tot[(tag, '_launches')]; dur = tot[(tag, '_dur_ns')]
        print("  pass %s: %d launches, avg %.1f us" % (tag, n, dur / max(n, 1) * 1e-3))
        for (t, k), v in sorted(tot.items()):
            if t == tag and not k.startswith('_'):
                print("     %-34s %.6g   (per launch %.6g)" % (k, v, v / max(n, 1)))
    g = lambda k, tag='p1': tot.get((tag, k), float('nan'))
    wc = g('SQ_WAVE_CYCLES')
    if wc == wc and wc > 0:
        print("  -- share of wave-cycles (pass p1):  parked (s_waitcnt / barrier) SQ_WAIT_ANY %.1f %%   issue-stalled SQ_WAIT_INST_ANY %.1f %% (of which LDS-issue %.1f %%)   "
              "issuing SQ_ACTIVE_INST_ANY %.1f %%" % (100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc, 100 * g('SQ_WAIT_INST_LDS') / wc, 100 * g('SQ_ACTIVE_INST_ANY') / wc))
        n = g('_launches'); dur = g('_dur_ns')
        print("  -- effective shader clock GRBM_GUI_ACTIVE / duration = %.3f GHz;  waves per launch %.0f" % (g('GRBM_GUI_ACTIVE') / dur, g('SQ_WAVES') / n))
    wc2 = tot.get(('p2', 'SQ_VALU_MFMA_BUSY_CYCLES'))
    if wc2:
        dur = tot[('p2', '_dur_ns')]
        print("  -- MFMA pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x time) = %.1f %%  (224 CUs of the masked stream = 87.5 %% of the SIMDs)" % (100 * wc2 / (1024 * 2.4 * dur)))
