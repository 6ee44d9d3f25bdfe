"""Config 5, two-phase SpMV cut into K row super-blocks (K virtual shards on the one GPU:
products of block k are written and read back before block k+1 starts -- do they stay in
the 256 MB Infinity Cache?).  Run under rocprofv3 --kernel-trace --stats; the figure of
merit is the summed k_pb_products + k_pb_reduce time per SpMV.  usage: gpu_pl_blocks.py K"""
import sys
sys.path.insert(0, ".")
import torch
import lsbench_amd as la

K = int(sys.argv[1])
la.hip_cdna4_init()
A = la.lsbench_matrix_synth("powerlaw:n=8000000,gamma=1.585350372615855,max=4096,seed=20240607")
s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, precond=la.PRECOND_NONE, spmv_variant=la.SPMV_TWOPHASE,
                                 nvirt=K, spmv_tune=0))
d_x = torch.sin(torch.arange(A.nrows, dtype=torch.float64, device="cuda"))
d_y = torch.empty_like(d_x)
for _ in range(12):
    s.spmv_dev(d_x, d_y)
torch.cuda.synchronize()
print("K", K, "done", float(d_y[:4].sum()))
s.destroy()
