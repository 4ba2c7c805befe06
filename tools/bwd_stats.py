"""Utilisation counters of render_bwd on the bench scene (config 3), default t_eps and t_eps = 0.
usage: bwd_stats.py [big]   (big = a small scene of huge faint splats: every batch overflows the slot pool)
Needs the stats build:  tools/build_variant.sh stats -DGS_STATS  &&  GSPLAT_LIB=$PWD/gaussian-splatting_amd/lib/var_stats.so python tools/bwd_stats.py"""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
dgr = importlib.import_module("diff_gaussian_rasterization")
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
L = importlib.import_module("gaussian-splatting_amd._lib")
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "big":     # the pool-overflow scene of tests/test_gpu_parity.py::test_midsize_vs_oracle
    sc = syn.make_scene(3000, 4, 100, 116, seed=6)
    sc.scales, sc.opacities = sc.scales * 12.0, sc.opacities * 0.03
else:
    sc = syn.make_scene(3_000_000, 16, 840, 1297, seed=0)
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
for t_eps in (1e-5, 0.0):
    st = dgr.GaussianRasterizationSettings(image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=sc.sh_degree, campos=sc.campos.to(dev), prefiltered=False, debug=False)
    img, radii, d, a = dgr.GaussianRasterizer(st, t_eps=t_eps)(means3D=p["means3D"], means2D=None, shs=p["shs"], colors_precomp=None, opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"], cov3D_precomp=None)
    torch.cuda.synchronize(); lib.gs_dbg_clear()
    img.backward(sc.G.to(dev)); torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)(); lib.gs_dbg_read(out); v = list(out)
    print(f"t_eps {t_eps}: wave-batches {v[0]}, pairs staged (x4 waves) {v[8]}, sum nmax {v[1]}, loop iterations {v[2]}, executed visits {v[3]}, "
          f"active lanes {v[4]} ({v[4]/max(v[3],1)/64:.3f} of the lanes of executed visits), live rows {v[5]} ({v[5]/max(v[3],1)/4:.3f}), "
          f"flush rounds {v[6]} (executed {v[7]}), valid flush lanes {v[9]} ({v[9]/max(v[7],1)/64:.3f}), "
          f"batches culled again for the slot pool (x4 waves) {v[10]}")
    if v[12]:
        print(f"   block-list entries {v[11]}; row-visits with the geometric block -> wave assignment {v[12]} (live {v[11]/v[12]:.3f}), with blocks "
              f"sorted by list length into groups of four {v[13]} (live {v[11]/max(v[13],1):.3f}, {v[13]/v[12]:.3f} of the visits), "
              f"blocks with a list packed in index order into as few waves as they need {v[14]} (live {v[11]/max(v[14],1):.3f}, {v[14]/v[12]:.3f} of the visits); waves with any list after sorting {v[15]} of {v[0]}")
