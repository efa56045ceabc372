"""Diagnostic (round 6): where does the diffusion head's bf16 FPN differ from the fp32 FPN on the same bf16 backbone maps?
Stage-wise relative L2 of (a) plain torch autocast bf16 (no padding, no fused kernels), (b) the product path (pad_to = 128, fused
top-down + folded lateral bias) against the fp32 FPN, per pyramid level; plus backbone run-to-run equality."""
import importlib, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
nnm = importlib.import_module("act3d-chained-diffuser_amd.nn")
import bench_denoise as BD
dev = torch.device("cuda:0")
m = BD.build_planner(a3d, dev, train=True)
head = m.prediction_head
s = BD.synthetic_inputs(2, 8, 2, dev)
x = s["rgbs"].flatten(0, 1)
with torch.no_grad():
    f1 = nnm.run_frozen_backbone(head.backbone, x, torch.bfloat16, keep_dtype=True, normalize=head.normalize)
    f1 = {k: v.clone() for k, v in f1.items()}
    f2 = nnm.run_frozen_backbone(head.backbone, x, torch.bfloat16, keep_dtype=True, normalize=head.normalize)
    for k in f1:
        print(f"backbone {k}: shape {tuple(f1[k].shape)} dtype {f1[k].dtype} cl {f1[k].is_contiguous(memory_format=torch.channels_last)} "
              f"run-to-run max diff {(f1[k].float() - f2[k].float()).abs().max().item():.3e} rms {f1[k].float().square().mean().sqrt().item():.3e} "
              f"absmax {f1[k].float().abs().max().item():.3e} finite {torch.isfinite(f1[k]).all().item()}")
    fpn = head.feature_pyramid
    E_ = fpn.inner_blocks[0][0].out_channels
    f32 = {k: v.float() for k, v in f1.items()}
    def rel(a, b):
        return ((a.float() - b.float()).norm() / b.float().norm()).item()
    for lvl in ("res5", "res4", "res3", "res2", "res1"):
        ref = fpn(f32, needed=[lvl])[lvl]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            plain = fpn(f1, needed=[lvl])[lvl]
            prod = fpn(f1, needed=[lvl], pad_to=(E_ + 63) // 64 * 64)[lvl]
        ref64 = fpn({k: v.double() for k, v in f32.items()}, needed=[lvl])[lvl] if False else None
        print(f"FPN out {lvl}: ref rms {ref.square().mean().sqrt().item():.3e} absmax {ref.abs().max().item():.3e} | plain autocast rel {rel(plain, ref):.3e} "
              f"| product (pad 128, fused) rel {rel(prod[:, :E_], ref):.3e} | product vs plain {rel(prod[:, :E_], plain):.3e} pad-channel absmax {prod[:, E_:].float().abs().max().item():.3e}")
    # the lateral convolutions alone
    for i, k in enumerate(f1):
        mconv = fpn.inner_blocks[i][0]
        ref = F.conv2d(f32[k], mconv.weight, mconv.bias)
        lo = F.conv2d(f1[k], mconv.weight.bfloat16(), mconv.bias.bfloat16())
        refd = F.conv2d(f32[k].double(), mconv.weight.double(), mconv.bias.double())
        print(f"lateral {k}: bf16 conv rel vs fp32 {rel(lo, ref):.3e}; fp32 conv vs float64 {rel(ref, refd):.3e}; bf16 vs float64 {rel(lo, refd):.3e}")
    mconv = fpn.layer_blocks[2][0]
    xin = torch.randn(4, E_, 32, 32, device=dev)
    refd = F.conv2d(xin.double(), mconv.weight.double(), mconv.bias.double(), padding=1)
    print(f"3x3 on randn: fp32 vs float64 {rel(F.conv2d(xin, mconv.weight, mconv.bias, padding=1), refd):.3e}; "
          f"bf16 vs float64 {rel(F.conv2d(xin.bfloat16(), mconv.weight.bfloat16(), mconv.bias.bfloat16(), padding=1), refd):.3e}")

# ---- which part of the backbone is not run-to-run reproducible at this 4-image batch?
def repro(tag, n_img=4, reps=3):
    xx = torch.rand(n_img, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    runs = []
    with torch.no_grad():
        for _ in range(reps):
            f = nnm.run_frozen_backbone(head.backbone, xx, torch.bfloat16, keep_dtype=True, normalize=head.normalize)
            runs.append({k: v.clone() for k, v in f.items()})
    d = {k: max((runs[0][k].float() - r[k].float()).abs().max().item() for r in runs[1:]) for k in runs[0]}
    print(f"repro [{tag}] N={n_img}: " + " ".join(f"{k}={v:.2e}" for k, v in d.items()))

repro("default")
repro("default", n_img=16)
repro("default", n_img=64)
for flag in ("FUSED_CONV1X1", "FUSED_CONV3X3", "FOLD_DOWNSAMPLE_BN", "FUSED_STEM"):
    old = getattr(nnm, flag); setattr(nnm, flag, False); repro(flag + "=0"); setattr(nnm, flag, old)
nnm.FUSED_CONV1X1 = False; nnm.FUSED_CONV3X3 = False; nnm.FUSED_STEM = False
repro("all convolutions MIOpen, fused BatchNorm")
nnm.FUSED_BN = False
repro("plain torch bf16 (MIOpen + torch BatchNorm)")
torch.backends.cudnn.deterministic = True
repro("plain torch bf16, cudnn.deterministic")
nnm.FUSED_BN = True
repro("all convolutions MIOpen, fused BatchNorm, cudnn.deterministic")
nnm.FUSED_CONV1X1 = True; nnm.FUSED_CONV3X3 = True; nnm.FUSED_STEM = True
repro("default, cudnn.deterministic")
repro("default, cudnn.deterministic", n_img=16)
