import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, importlib
import diff_gaussian_rasterization as dgr
from oracle import oracle
syn = importlib.import_module("gaussian-splatting_amd.synthetic")
dev = torch.device("cuda:0")
H = W = 16
sc = syn.make_scene(1, 1, H, W, seed=0)
sc.means3D = torch.tensor([[float(sys.argv[1]) if len(sys.argv) > 1 else -0.2, -0.2, 4.0]])
sc.scales = torch.tensor([[0.15, 0.1, 0.1]]); sc.opacities = torch.tensor([[0.7]])
G_C = torch.ones(3, H, W) * torch.tensor([1.0, 2.0, 3.0]).view(3, 1, 1); G_Z = torch.ones(1, H, W) * 0.5; G_A = torch.ones(1, H, W) * 0.25
a = lambda t: t.numpy()
geom = oracle.preprocess(H, W, 0, a(sc.viewmatrix), a(sc.projmatrix), a(sc.campos), a(sc.means3D), scales=a(sc.scales), rotations=a(sc.rotations), shs=a(sc.shs))
print("rect", geom.rect, "xy", geom.xy)
rb = oracle.render_backward(geom, a(sc.opacities), a(G_C), a(G_Z), a(G_A))
ref = np.concatenate([rb["d_xy"], rb["d_conic"], rb["d_opacity"], rb["d_rgb"], rb["d_depth"][:, None]], axis=1)
st = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=1, tanfovy=1, bg=sc.bg.to(dev), scale_modifier=1.0, viewmatrix=sc.viewmatrix.to(dev), projmatrix=sc.projmatrix.to(dev), sh_degree=0, campos=sc.campos.to(dev), prefiltered=False, debug=False)
R = dgr._impl.rasterizer; lib = dgr._impl._lib.load(); p = R._ptr; s = R._stream()
bs = R.preprocess_and_bin(st, sc.means3D.to(dev), sc.opacities.to(dev), sc.scales.to(dev), sc.rotations.to(dev), None, sc.shs.to(dev), None)
img = torch.empty(3, H, W, device=dev); dep = torch.empty(1, H, W, device=dev); alp = torch.empty(1, H, W, device=dev); aux = torch.empty(3, H, W, device=dev)
bg = sc.bg.to(dev)
assert lib.gs_render_fwd(H, W, p(bg), p(bs.splat), p(bs.ranges), p(bs.point_list), p(img), p(dep), p(alp), p(aux), 0.0, 0, 1, s) == 0
rows = torch.zeros(1, 16, device=dev)
gc, gz, ga = G_C.to(dev), G_Z.to(dev), G_A.to(dev)
assert lib.gs_render_bwd(H, W, p(bs.splat), p(bs.ranges), p(bs.point_list), p(aux), p(dep), p(alp), p(gc), p(gz), p(ga), p(rows), 0.0, 0, 1, s) == 0
torch.cuda.synchronize()
np.set_printoptions(precision=5, suppress=True, linewidth=200)
print("got", rows.cpu().numpy()[0, :12]); print("ref", ref[0])
