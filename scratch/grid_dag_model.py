"""Event model of the partitioned factorisation on W ranks: the dependency graph of gptools_amd.dist (GridLML's three queues per
rank, DistributedLML's two) walked with MEASURED operation times.  Pure Python, no GPU: the per-term budget of DESIGN.md
section 5 and its sensitivities (what the modelled 8-rank time would be if the chain ran alone on the chip, at its in-situ
cost, with a free interconnect, ...).

Inputs (one idle MI355X, profiles/r04_chain_terms.txt; bench / replay traces for the rest):
  diagonal block 512: update 12 + factor 108 + inverse 69 us;  512^3 product 12 us;
  m x 512 x 512 GEMM (slices, look-ahead updates): 31 / 40 / 50 / 54 / 56 TFLOP/s at m = 1 / 2 / 4 / 8 / 16 k rows;
  trailing update of a rank: its flops at 50 TFLOP/s (staircase launches in situ on 224 CUs: 45-55) + 20 us per launch;
  links: bytes / 153 GB/s + latency per hop (xGMI, point to point; a root feeds its peers over different links at once).
`slow` multiplies every CHAIN and BULK kernel (they share the chip with a trailing update while one is running: replay traces
show 0.45-0.77 ms for the 0.19 ms diagonal block, i.e. slow = 2.4-4).

  python scratch/grid_dag_model.py [N] [nb]
"""
import sys
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NP = (N + 1 + nb - 1) // nb * nb
nblk = NP // nb
BW = 153e9
T_DIAG = (12 + 108 + 69) * 1e-6 * (nb / 512.0)
T_FACT = 108e-6 * (nb / 512.0)
T_INV = 69e-6 * (nb / 512.0)
T_G = 12e-6 * (nb / 512.0) ** 2
UPD_RATE, UPD_LAUNCH = 50e12, 20e-6


def gemm_rows(m):
    """m x nb x nb product on the panel / bulk queue (measured rates, interpolated in log m)."""
    if m <= 0:
        return 0.0
    ms = np.array([1024, 2048, 4096, 8192, 16384], float)
    rs = np.array([30.9, 39.7, 49.5, 54.1, 55.7]) * 1e12
    r = np.interp(np.log(max(m, 1024)), np.log(ms), rs)
    return max(2.0 * m * nb * nb / r, T_G)


def hop(nbytes, lat):
    return lat + nbytes / BW


def model_grid(Pr, Pc, lat, slow, free_links=False, verbose=False):
    W = Pr * Pc
    bw_t = (lambda b: 0.0) if free_links else (lambda b: b / BW)
    lat_ = 0.0 if free_links else lat
    rows = {r: [I for I in range(nblk) if I % Pr == r] for r in range(Pr)}
    cols = {c: [J for J in range(nblk) if J % Pc == c] for c in range(Pc)}
    # queue-free times per rank
    qc = np.zeros((Pr, Pc)); qb = np.zeros((Pr, Pc)); qm = np.zeros((Pr, Pc))
    t_W = {}; t_H = {}; t_R0 = {}; t_R = {}; t_C = {}        # arrival times per (k, rank)
    urg = {}                                                # (k, rank): urgent part of update k done
    done = {}
    acc = dict(chain=0.0)
    INF = 0.0

    def nrows_ge(r, I0):
        return sum(1 for I in rows[r] if I >= I0)

    def ncols_ge(c, J0):
        return sum(1 for J in cols[c] if J >= J0)

    last_chain_t = 0.0
    chain_at = []
    for p in range(nblk):
        k = p - 1
        pcp, prp = p % Pc, p % Pr
        # ---- chain: diagonal owner
        D = (prp, pcp)
        dep = qc[D]
        if k >= 0:
            dep = max(dep, t_H[(k, D)], urg.get((k - 1, D), 0.0))
        t_fact = dep + slow * (T_DIAG if k >= 0 else T_FACT + T_INV)
        qc[D] = t_fact
        for r in range(Pr):
            t_W[(p, (r, pcp))] = t_fact + (0.0 if r == prp else hop(8.0 * nb * nb, lat_) if not free_links else 0.0)
        # ---- chain: head block on ((p+1)%Pr, pcp)
        if p + 1 < nblk:
            Hd = ((p + 1) % Pr, pcp)
            dep = qc[Hd]
            if k >= 0:
                dep = max(dep, t_H[(k, Hd)], t_R0.get((k, Hd), 0.0), urg.get((k - 1, Hd), 0.0))
                dep += slow * T_G
            dep = max(dep, t_W[(p, Hd)]) + slow * T_G
            qc[Hd] = dep
            for r in range(Pr):
                for c in range(Pc):
                    t_H[(p, (r, c))] = dep + (0.0 if (r, c) == Hd else (0.0 if free_links else hop(8.0 * nb * nb, lat_)))
            chain_at.append(dep)
        # ---- bulk on every rank of process column pcp
        for r in range(Pr):
            me = (r, pcp)
            t = qb[me]
            if k >= 0:
                lu_rows = nrows_ge(r, p + 1) - (1 if r == (p + 1) % Pr else 0)
                if lu_rows > 0:
                    t = max(t, t_H[(k, me)], t_R0.get((k, me), 0.0), t_R.get((k, me), 0.0), urg.get((k - 1, me), 0.0))
                    t += slow * gemm_rows(lu_rows * nb)
            m = nrows_ge(r, p + 2)
            has0 = (r == (p + 2) % Pr) and p + 2 < nblk and m > 0
            if m > 0:
                t = max(t, t_W[(p, me)])
            if has0:
                t += slow * T_G
                for c in range(Pc):
                    t_R0[(p, (r, c))] = t + (0.0 if c == pcp or free_links else hop(8.0 * nb * nb, lat_))
            rest = m - (1 if has0 else 0)
            if rest > 0:
                t += slow * gemm_rows(rest * nb)
            for c in range(Pc):
                t_R[(p, (r, c))] = t + (0.0 if c == pcp or free_links or rest <= 0 else hop(8.0 * rest * nb * nb, lat_))
            qb[me] = t
        # other ranks: R arrives by broadcast (set above); exchange on every rank's bulk queue
        for r in range(Pr):
            for c in range(Pc):
                me = (r, c)
                ncol = ncols_ge(c, p + 2)
                if ncol <= 0:
                    t_C[(p, me)] = 0.0
                    continue
                # pieces come from the process rows that hold the rank's columns: ready when THEIR R is complete
                srcs = sorted(set(J % Pr for J in cols[c] if J >= p + 2))
                t = max(qb[me], t_R0.get((p, me), 0.0), t_R.get((p, me), 0.0))
                for q in srcs:
                    nq = sum(1 for J in cols[c] if J >= p + 2 and J % Pr == q)
                    ready = max(t_R0.get((p, (q, c)), 0.0), t_R.get((p, (q, c)), 0.0))
                    t = max(t, ready) + (0.0 if free_links or Pr == 1 else hop(8.0 * nq * nb * nb, lat_)) + 8.0 * nq * nb * nb * 2 / 2.0e12
                t_C[(p, me)] = t
                qb[me] = t
        # ---- main queue: update p on every rank
        for r in range(Pr):
            for c in range(Pc):
                me = (r, c)
                nblocks = sum(1 for J in cols[c] if J >= p + 2 for I in rows[r] if I >= J)
                t = max(qm[me], t_R0.get((p, me), 0.0), t_R.get((p, me), 0.0), t_C[(p, me)])
                ucol = (p + 2) % Pc == c and any(I >= p + 2 for I in rows[r])
                if nblocks > 0:
                    nurg = sum(1 for I in rows[r] if I >= p + 2) if ucol else 0
                    tu = t + (UPD_LAUNCH + 2.0 * nurg * nb ** 3 / UPD_RATE if nurg else 0.0)
                    urg[(p, me)] = tu
                    t = tu + UPD_LAUNCH + 2.0 * (nblocks - nurg) * nb ** 3 / UPD_RATE
                else:
                    urg[(p, me)] = t
                qm[me] = t
                done[(p, me)] = t
    T = float(qm.max())
    return T, chain_at


def model_1d(Wn, lat, slow, sag=True):
    """DistributedLML: per panel the owner applies panel k-1 to column k (m x nb x nb), factors (diagonal block + inverse + one
    GEMM of the rows below) and moves the whole panel (scatter + all-gather: 2 bytes / (W bw) + 2 latencies)."""
    qp = np.zeros(Wn); qm = np.zeros(Wn)
    arr = {}
    urg = {}
    for p in range(nblk):
        o = p % Wn
        m = (nblk - p) * nb
        t = qp[o]
        if p > 0:
            t = max(t, arr[p - 1][o], urg.get((p - 2, o), 0.0)) + slow * gemm_rows(m)
        t += slow * (T_FACT + T_INV + gemm_rows(m - nb))
        qp[o] = t
        nbytes = 8.0 * m * nb
        tt = (2.0 * nbytes / (Wn * BW) + 2 * lat) if sag else (nbytes / BW + lat)
        arr[p] = [t if r == o else t + tt for r in range(Wn)]
        for r in range(Wn):
            mine = [J for J in range(p + 1, nblk) if J % Wn == r and J != p + 1]
            t0 = max(qm[r], arr[p][r])
            fl = sum(2.0 * nb * nb * ((nblk - J) * nb - nb / 2.0) for J in mine)
            u = p + 2
            if u in mine:
                fu = 2.0 * nb * nb * ((nblk - u) * nb - nb / 2.0)
                urg[(p, r)] = t0 + UPD_LAUNCH + fu / UPD_RATE
                fl -= fu
                t0 = urg[(p, r)]
            qm[r] = t0 + (UPD_LAUNCH + fl / UPD_RATE if fl > 0 else 0.0)
    return float(qm.max())


flops = N ** 3 / 3.0 + N ** 2 / 2.0 + N / 6.0 + 2.0 * N ** 2
print("# event model of the partitioned factorisation, N = %d, nb = %d (%d panels), 8 ranks; target 50 %% of 8 x 78.6 TFLOP/s = %.1f ms" % (
    N, nb, nblk, flops / (0.5 * 8 * 78.6e12) * 1e3))
print("# slow = factor on every chain / bulk kernel (1: alone on the chip; 2.4-4: beside a trailing update, replay traces)")
print("%-34s %8s %8s %8s %8s" % ("layout, links", "slow=1", "slow=2", "slow=3", "slow=4"))
for name, fn in (("1-D, scatter+all-gather, 20 us", lambda s: model_1d(8, 20e-6, s)),
                 ("1-D, broadcast one link, 20 us", lambda s: model_1d(8, 20e-6, s, sag=False)),
                 ("grid 2x4, 20 us", lambda s: model_grid(2, 4, 20e-6, s)[0]),
                 ("grid 4x2, 20 us", lambda s: model_grid(4, 2, 20e-6, s)[0]),
                 ("grid 2x4, 50 us", lambda s: model_grid(2, 4, 50e-6, s)[0]),
                 ("grid 2x4, free interconnect", lambda s: model_grid(2, 4, 0.0, s, free_links=True)[0]),
                 ("grid 8x1 (block rows), 20 us", lambda s: model_grid(8, 1, 20e-6, s)[0]),
                 ("grid 1x8 (= 1-D by this engine)", lambda s: model_grid(1, 8, 20e-6, s)[0])):
    ts = [fn(s) * 1e3 for s in (1.0, 2.0, 3.0, 4.0)]
    print("%-34s %8.1f %8.1f %8.1f %8.1f   ms   (%s %% of peak)" % (name, ts[0], ts[1], ts[2], ts[3],
                                                                   " / ".join("%.0f" % (100 * flops / (t * 1e-3) / (8 * 78.6e12)) for t in ts)))
T, chain = model_grid(2, 4, 20e-6, 1.0)
d = np.diff(np.array(chain))
q_ = [0, len(chain) // 4, len(chain) // 2, 3 * len(chain) // 4, len(chain) - 1]
print("# grid 2x4, slow = 1: head block of panel p ready at (ms): " + ", ".join("p=%d %.2f" % (i, chain[i] * 1e3) for i in q_) +
      "; chain period in the chain-bound tail (last quarter of the panels): %.3f ms" % (d[-len(chain) // 4:].mean() * 1e3))
upd = sum(2.0 * nb ** 3 * sum(1 for J in range(p + 2, nblk) for I in range(J, nblk)) for p in range(nblk)) / 8 / UPD_RATE
print("# per-rank trailing updates at %.0f TFLOP/s: %.1f ms; diagonal-block work alone: %d x %.0f us = %.1f ms" % (
    UPD_RATE * 1e-12, upd * 1e3, nblk, T_DIAG * 1e6, nblk * T_DIAG * 1e3))
