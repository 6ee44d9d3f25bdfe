"""Which layout of the 16-bit sliced-ELL form differs from which, and where (diagnostic)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import lsbench_amd as hip
from oracle import oracle as O
hip.hip_cdna4_init()
for spec in sys.argv[1:] or ["lap3d:nx=64,ny=64,nz=40"]:
    A = hip.lsbench_matrix_synth(spec)
    xs = np.sin(np.arange(A.nrows, dtype=np.float64))
    out = {}
    for name, off, tune in (("full", "1", 6), ("const", None, 6), ("tmpl", None, 70), ("sell32", None, 2)):
        if off:
            os.environ["LSBENCH_HIP_NO_VCONST"] = off
        else:
            os.environ.pop("LSBENCH_HIP_NO_VCONST", None)
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, spmv_variant=hip.SPMV_SELL, spmv_tune=tune, use_graph=0))
        d_y = torch.empty(A.nrows, dtype=torch.float64, device="cuda:0")
        s.spmv_dev(torch.from_numpy(xs).to("cuda:0"), d_y)
        out[name] = d_y.cpu().numpy()
        print(spec, name, "flags", s.spmv_flags, "slots", s.sell_value_slots, "bytes", s.spmv_layout_bytes)
        s.destroy()
    for a in ("const", "tmpl", "sell32"):
        d = np.nonzero(out[a] != out["full"])[0]
        print(a, "vs full: differing rows", len(d), d[:12], (out[a][d[:4]] - out["full"][d[:4]]) if len(d) else "")
