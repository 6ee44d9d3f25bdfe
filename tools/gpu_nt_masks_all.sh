cd "$GRAFT_REPO_ROOT"
echo "== lap2d more masks"; MASKS="5 9 13 11 4 12 2 6 10 14" bash tools/gpu_nt_masks.sh r3_nt2 lap2d 1500
echo "== lap2d_coef"; MASKS="31 5 9 13 0" bash tools/gpu_nt_masks.sh r3_nt2c lap2d_coef 800
echo "== lap3d"; MASKS="31 5 9 13 0" bash tools/gpu_nt_masks.sh r3_nt23 lap3d 150
echo "== 1/8 of config 3"; MASKS="31 5 9 13 0" bash tools/gpu_nt_masks.sh r3_nt28 lap2d:nx=3162,ny=395 4000
