"""match_union8_kernel scores every (row, query) cell of a pass of 8 rows x 8 queries; 45 % of the cells are not members of
the query's radius.  How many (pass, query PAIR) blocks of 8 SAD instructions have no member at all -- in the kernel's list
order (y buckets) and with the list ordered by which pairs a row belongs to -- on the bench's geometry (numpy model of the
tile / round / list composition, no GPU):  python3 tools/experiments/union_skip_estimate.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from libviso_amd import synth
s = synth.make_sequence(1000, 3, n_kp=2000)
R = 80.0; K = 250
rowpairs=0; emptyrp=0; eb2=0; blocks2=0; tot_cells=0; mem_cells=0; blocks=0; empty_blocks=0; empty_q=0; passes=0
for (a,b) in ((1,0),(2,1)):
  for side in (0,1):
    q = s["kp"][a, side, :s["n"][a, side]]; t = s["kp"][b, side, :s["n"][b, side]]
    # column buckets: 256 buckets over x; order by bucket (stable), tiles of 64 consecutive
    x0, x1 = q[:,0].min(), q[:,0].max()
    bq = np.minimum(((q[:,0]-x0)*(256/(x1-x0))).astype(int), 255)
    order = np.argsort(bq, kind='stable')
    qs = q[order]
    for t0 in range(0, len(qs), 64):
        tile = qs[t0:t0+64]
        yo = np.argsort(tile[:,1], kind='stable')
        tile = tile[yo]
        for r0 in range(0, len(tile), 8):
            rq = tile[r0:r0+8]
            d = np.abs(rq[:,None,0]-t[None,:,0]) + np.abs(rq[:,None,1]-t[None,:,1])   # [8][n2]
            mem = d <= R
            # K cap: nearest K only (approx)
            for k in range(len(rq)):
                if mem[k].sum() > K:
                    thr = np.sort(d[k][mem[k]])[K-1]
                    mem[k] &= d[k] <= thr
            inu = mem.any(0)
            idx = np.nonzero(inu)[0]
            # list order: y buckets (64 over the target's y range), inside a bucket window order (x bucket order ~ x)
            ty = t[idx,1]; yb = np.minimum(((ty - t[:,1].min())*(64/(t[:,1].max()-t[:,1].min()))).astype(int), 63)
            o = np.lexsort((t[idx,0], yb))
            idx = idx[o]
            m = mem[:, idx]
            if m.shape[0] < 8: m = np.vstack([m, np.zeros((8-m.shape[0], m.shape[1]), bool)])
            n = m.shape[1]
            pm = m.reshape(4,2,n).any(1)            # [4][n]: the row belongs to a query of pair J
            rowpairs += pm.size; emptyrp += (~pm).sum()
            sig = (pm * np.array([[1],[2],[4],[8]])).sum(0)
            so = np.argsort(sig ^ (sig >> 1), kind='stable')   # rows ordered by their pair signature (Gray order)
            pms = pm[:, so]
            for p0 in range(0, n, 8):
                blk = pms[:, p0:p0+8]
                for pair in range(4):
                    blocks2 += 1
                    if not blk[pair].any(): eb2 += 1
            tot_cells += 8*n; mem_cells += m.sum()
            for p0 in range(0, n, 8):
                mp = m[:, p0:p0+8]
                passes += 1
                for pair in range(4):
                    blocks += 1
                    if not mp[2*pair:2*pair+2].any(): empty_blocks += 1
                for k in range(8):
                    if not mp[k].any(): empty_q += 1
print("(row, pair) without a member: %.1f %% (what perfect grouping could skip); lists ordered by pair signature: %.1f %% of the (pass, pair) blocks empty" % (100*emptyrp/rowpairs, 100*eb2/blocks2)); print("member cells %.1f %%; (pass, query pair) blocks without a member: %.1f %%; (pass, query) without a member: %.1f %%; rows per round %.1f" % (100*mem_cells/tot_cells, 100*empty_blocks/blocks, 100*empty_q/(8*passes), tot_cells/8/ (passes/ (1)) *0 + tot_cells/8/max(1,passes)*1))
