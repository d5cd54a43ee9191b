"""Per-layer cost table of the ResNetV2 stem convolutions (bf16, 32 views of 448^2) on MIOpen: fwd / bwd per distinct shape."""
import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd.backbone import ResNetV2, StdConv2dSame
import torch.nn.functional as F
dev = "cuda:0"
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
net = ResNetV2().to(dev).bfloat16()
shapes = collections.OrderedDict()
hooks = []
def mk(name):
    def hook(m, inp, out):
        x = inp[0]
        key = (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], x.shape[-2], x.shape[-1])
        shapes[key] = shapes.get(key, 0) + 1
    return hook
for n, m in net.named_modules():
    if isinstance(m, StdConv2dSame): hooks.append(m.register_forward_hook(mk(n)))
x = torch.randn(32, 3, 448, 448, device=dev).bfloat16()
with torch.no_grad(): net(x)
for h in hooks: h.remove()
tot_f = tot_b = 0
print("cin cout k s  HxW   count   fwd us   bwd us   GF(fwd)  TF/s fwd  TF/s bwd")
for (cin, cout, k, s, H, W), cnt in shapes.items():
    xi = torch.randn(32, cin, H, W, device=dev).bfloat16().requires_grad_(True)
    w = torch.randn(cout, cin, k, k, device=dev).bfloat16().requires_grad_(True)
    conv = StdConv2dSame(cin, cout, k, stride=s).to(dev).bfloat16()
    def f():
        xp = xi
        if conv.dynamic_pad:
            from acr_wsss_amd.backbone import pad_same
            xp = pad_same(xi, k, s)
        return F.conv2d(xp, w, None, s, conv.padding)
    y = f(); dy = torch.randn_like(y)
    tf = t(lambda: f())
    def fb():
        y = f(); y.backward(dy); xi.grad = None; w.grad = None
    tb = t(fb) - tf
    gf = 2.0 * 32 * y.shape[-1] * y.shape[-2] * cout * cin * k * k / 1e9
    print("%4d %4d %d %d %3dx%-3d  x%2d  %8.1f %8.1f  %7.1f  %7.1f  %7.1f" % (cin, cout, k, s, H, W, cnt, tf, tb, gf, gf / tf / 1e3, 2 * gf / tb / 1e3))
    tot_f += cnt * tf; tot_b += cnt * tb
print("total conv fwd %.2f ms  bwd %.2f ms" % (tot_f / 1e3, tot_b / 1e3))
