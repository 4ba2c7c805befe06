"""A/B of rasterizer.SKIP_CULLED_RANKS (gs_scan_tiles with / without the sorted depth keys) on the bench scene: scan_tiles stage ms.
usage: python tools/ab_scan_skip.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
R = importlib.import_module("gaussian-splatting_amd.rasterizer")
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
dgr = importlib.import_module("diff_gaussian_rasterization")
dev = torch.device("cuda:0")
sc = syn.make_scene(3_000_000, 16, 840, 1297, seed=0)
st = dgr.GaussianRasterizationSettings(image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=sc.sh_degree, campos=sc.campos.to(dev), prefiltered=False, debug=False)
p = {k: getattr(sc, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
def run(n):
    for _ in range(n):
        s = R.preprocess_and_bin(st, p["means3D"], p["opacities"], p["scales"], p["rotations"], None, p["shs"], None)
    torch.cuda.synchronize()
    return s
for rep in range(2):
    for flag in (False, True):
        R.SKIP_CULLED_RANKS = flag
        run(5)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        # time the whole binning chain (the scan is inside it), 30 frames
        e[0].record(); s = run(30); e[1].record(); torch.cuda.synchronize()
        print(f"skip={flag}: preprocess+binning {e[0].elapsed_time(e[1]) / 30:.4f} ms per frame, D = {s.num_pairs}")
