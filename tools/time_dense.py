"""BASELINE config c2: one large dense QP (n=1024, m=2048, 307 strongly active) through Mode A; times the GPU
solve and the CPU oracle on the same problem and checks bit-exact parity."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
from oracle_lib import get_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
t = time.time(); pb = P.generate_batch(1, P.benchmark_characteristics(n, m), first_index=0); print(f"generated n={n} m={m} ma={int(0.3*n)} in {time.time()-t:.1f}s", flush=True)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
a, C, bl, bu = dev(pb.a[0]), dev(pb.C[0]), dev(pb.bl[0]), dev(pb.bu[0])
s = capi.DenseSolver(n, m, False, max_iter=100000)
times = []
for rep in range(3):
    G = dev(pb.G[0]); torch.cuda.synchronize()
    t = time.perf_counter(); r = s.solve(G, a, C, bl, bu); torch.cuda.synchronize(); times.append(time.perf_counter() - t)
print(f"GPU: term {r['term']} iters {r['iters']} f {r['f']:.6e}  wall {min(times)*1e3:.1f} ms (best of 3: {[f'{x*1e3:.1f}' for x in times]})  err vs planted {np.linalg.norm(r['x']-pb.x_star[0]):.2e}", flush=True)
o = get_oracle()
t = time.perf_counter(); ref = o.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=100000); tc = time.perf_counter() - t
print(f"CPU oracle (1 thread): term {ref.term} iters {ref.iters} wall {tc*1e3:.1f} ms")
same = ref.term == r["term"] and ref.iters == r["iters"] and np.array_equal(ref.status, r["status"]) and np.array_equal(ref.x, r["x"]) and np.array_equal(ref.u, r["u"]) and ref.f == r["f"]
print("bit-identical to the oracle:", same, " speed-up vs 1 CPU thread:", f"{tc/min(times):.1f}x")
