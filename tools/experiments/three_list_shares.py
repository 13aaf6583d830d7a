"""match_union8_kernel: share of a round's union rows whose members lie in one half of the round only (queries 0-3 / 4-7), and what\nhalf passes for those rows would save (sized without a GPU):  python3 tools/experiments/three_list_shares.py"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from libviso_amd import synth
def run(cluster):
    s = synth.make_sequence(1000, 3, n_kp=2000, cluster_frac=cluster)
    R = 80.0
    tot_rows=0; nA=nB=nAB=0; cost_now=0; cost_new=0; cost_new_q=0; rounds=0; memcells=0
    cost_quart=0
    for (a,b) in ((1,0),(2,1)):
      for side in (0,1):
        q = s["kp"][a, side, :s["n"][a, side]]; t = s["kp"][b, side, :s["n"][b, side]]
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
                d = np.abs(rq[:,None,0]-t[None,:,0]) + np.abs(rq[:,None,1]-t[None,:,1])
                mem = d <= R
                if mem.shape[0] < 8: mem = np.vstack([mem, np.zeros((8-mem.shape[0], mem.shape[1]), bool)])
                inu = mem.any(0)
                m = mem[:, inu]
                n = m.shape[1]
                a_ = m[:4].any(0); b_ = m[4:].any(0)
                A = (a_ & ~b_).sum(); B = (b_ & ~a_).sum(); AB = (a_ & b_).sum()
                nA += A; nB += B; nAB += AB; tot_rows += n; rounds += 1; memcells += m.sum()
                cost_now += -(-n//8)*32
                cost_new += -(-A//16)*32 + -(-B//16)*32 + -(-AB//8)*32
                # pair-granular: 4 pairs; rows classified by which pairs they belong to -> ideal = sum over pairs of ceil(rows_in_pair/8)*8
                pm = m.reshape(4,2,n).any(1)
                cost_quart += sum(-(-int(pm[j].sum())//8)*8 for j in range(4))
    print("cluster %.1f: rows/round %.1f  A-only %.1f%% B-only %.1f%% both %.1f%%  member cells %.1f%%" % (cluster, tot_rows/rounds, 100*nA/tot_rows, 100*nB/tot_rows, 100*nAB/tot_rows, 100*memcells/(8*tot_rows)))
    print("   SAD instr per round now %.1f, three-list %.1f (%.1f%% fewer), per-pair lists ideal %.1f (%.1f%% fewer)" % (cost_now/rounds, cost_new/rounds, 100*(1-cost_new/cost_now), cost_quart/rounds, 100*(1-cost_quart/cost_now)))
run(0.0); run(0.7)
