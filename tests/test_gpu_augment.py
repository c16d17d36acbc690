"""Float augmentations on the MI355X against the oracle restatements (ColorJiggle: parity unpinned to the
reference, pinned to this build's own statement; NoisyCover cover logic pinned by noisy_cover.npz), and the
float-NHWC source of the first conv against the NCHW tensor contract."""
import numpy as np
import pytest
import torch

from tests._util import RTOL, load, rel_err

pytestmark = pytest.mark.gpu


def _ring(frames_chw):
    n, c, h, w = frames_chw.shape
    store = torch.zeros(n * c * h * w + 32, dtype=torch.uint8, device="cuda")
    ring = store[:n * c * h * w].view(n, h, w, c)
    ring.copy_(torch.from_numpy(frames_chw).permute(0, 2, 3, 1))
    return ring


@pytest.mark.parametrize("C,H,W,B", [(9, 34, 40, 6), (12, 21, 19, 5), (3, 8, 8, 2)])
def test_color_jiggle_kernel_vs_restatement(C, H, W, B):
    from curla_amd import ops
    from oracle import curla_oracle as O
    rs = np.random.RandomState(C)
    frames = rs.randint(0, 256, (10, C, H, W), dtype=np.uint8)
    ring = _ring(frames)
    idx = rs.randint(0, 10, B)
    k = C // 3
    torch.manual_seed(C)
    params = torch.stack([(torch.rand(B * k) < 0.7).float(), torch.empty(B * k).uniform_(0.8, 1.2),
                          torch.empty(B * k).uniform_(0.5, 1.5), torch.empty(B * k).uniform_(-3.14, 3.14)], 1).contiguous()
    for order in ([0, 1, 2, 3], [3, 1, 0, 2], [2, 3, 1, 0]):
        ref = O.color_jiggle(frames[idx], params, order)
        out = torch.full((B, H, W, C), float("nan"), device="cuda")
        ops.color_jiggle(ring, torch.from_numpy(idx).cuda(), params.cuda(), torch.tensor(order, dtype=torch.int32).cuda(), B, out)
        got = out.permute(0, 3, 1, 2).cpu()
        # hue sectors are discontinuous in the intermediate (h, f) but continuous in RGB: compare in RGB
        assert rel_err(got, ref) <= 1e-4, (order, rel_err(got, ref))


def test_noisy_cover_kernel_vs_reference_fixture():
    from curla_amd import ops
    g = load("noisy_cover.npz")
    rs = np.random.RandomState(int(g["imgs_seed"]))
    imgs = rs.randint(0, 256, (5, 9, 34, 40), dtype=np.uint8)
    noise = rs.randn(5, 9, 34, 40).astype(np.float32) * 10.0
    ring = _ring(imgs)
    out = torch.full((5, 34, 40, 9), float("nan"), device="cuda")
    ops.noisy_cover(ring, None, torch.from_numpy(noise).permute(0, 2, 3, 1).contiguous().cuda(), list(g["colors"]),
                    int(g["top"]), int(g["bottom"]), 5, out)
    got = out.permute(0, 3, 1, 2).cpu()
    assert abs(got.double().sum().item() - float(g["out_sum"])) <= 1e-6 * abs(float(g["out_sum"]))
    assert np.abs(got.numpy() - g["out"].astype(np.float32)).max() <= 0.13


def test_augmentor_tensor_api_color_jiggle():
    """ColorJiggle.training_augmentation(batch) -- the reference's own call (augmentations.py:105-136,
    utils.py:174-182) -- runs the jitter kernel on a float NCHW tensor: equal to the restatement, and bit-equal
    to what ReplayBuffer produces from the ring with the same parameters."""
    import curla_amd
    from curla_amd import ops
    from oracle import curla_oracle as O
    C, H, W, B = 12, 26, 30, 5
    rs = np.random.RandomState(4)
    frames = rs.randint(0, 256, (B, C, H, W), dtype=np.uint8)
    aug = curla_amd.make_augmentor("color_jiggle", (H, W))
    torch.manual_seed(2)
    params, order = aug.draw_params(B * (C // 3))
    x = torch.from_numpy(frames).float().cuda()
    keep = x.clone()
    got = aug.training_augmentation(x, params=params, order=order)
    assert got.shape == x.shape and got.dtype == torch.float32 and got.is_cuda
    assert torch.equal(x, keep)  # the argument is left alone
    assert rel_err(got.cpu(), O.color_jiggle(frames, params, order)) <= 1e-4
    nhwc = torch.empty((B, H, W, C), device="cuda")
    ops.color_jiggle(_ring(frames), None, params.cuda(), order.cuda(), B, nhwc)
    assert torch.equal(got, nhwc.permute(0, 3, 1, 2).contiguous())
    # with its own random draws: in range, changed, frames jittered independently
    torch.manual_seed(3)
    rnd = aug.training_augmentation(x)
    assert float(rnd.min()) >= 0 and float(rnd.max()) <= 255.001 and not torch.equal(rnd, x)
    with pytest.raises(ValueError):
        aug.training_augmentation(x[:, :, :-1])
    with pytest.raises(RuntimeError):
        aug.training_augmentation(x.cpu())


def test_augmentor_tensor_api_noisy_cover():
    """NoisyCover.training_augmentation(batch) (augmentations.py:170-205) against the reference-generated fixture."""
    import curla_amd
    g = load("noisy_cover.npz")
    rs = np.random.RandomState(int(g["imgs_seed"]))
    imgs = rs.randint(0, 256, (5, 9, 34, 40), dtype=np.uint8)
    noise = rs.randn(5, 9, 34, 40).astype(np.float32) * 10.0
    aug = curla_amd.make_augmentor("noisy_cover", (34, 40))
    assert (aug.top, aug.bottom) == (int(g["top"]), int(g["bottom"]))
    x = torch.from_numpy(imgs).float().cuda()
    got = aug.training_augmentation(x, colors=list(g["colors"]), noise=torch.from_numpy(noise).cuda()).cpu()
    assert abs(got.double().sum().item() - float(g["out_sum"])) <= 1e-6 * abs(float(g["out_sum"]))
    assert np.abs(got.numpy() - g["out"].astype(np.float32)).max() <= 0.13
    np.random.seed(0)
    rnd = aug.training_augmentation(x)  # own draws: 3 colours from NumPy's stream, device Gaussian noise
    np.random.seed(0)
    cols = [np.random.randint(0, 255) for _ in range(3)]
    assert abs(float(rnd[:, 0, :aug.top].mean()) - cols[0]) < 3.0 and float(rnd.max()) <= 255.0


def test_gather_and_conv1_nhwc_source_matches_nchw_contract():
    from curla_amd import ops
    rs = np.random.RandomState(1)
    C, H, W, B = 12, 41, 37, 4
    frames = rs.randint(0, 256, (7, C, H, W), dtype=np.uint8)
    ring = _ring(frames)
    idx = torch.from_numpy(rs.randint(0, 7, B)).cuda()
    nhwc = torch.empty((B, H, W, C), device="cuda")
    ops.gather_nhwc(ring, idx, B, nhwc)
    assert np.array_equal(nhwc.permute(0, 3, 1, 2).cpu().numpy(), frames[idx.cpu().numpy()].astype(np.float32))
    nhwc += 0.37  # non-integer pixels, as after colour jitter
    nchw = nhwc.permute(0, 3, 1, 2).contiguous()
    w = torch.randn(32, C, 3, 3, device="cuda") * 0.2
    b = torch.randn(32, device="cuda") * 0.1
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    o1 = torch.empty(B, Ho, Wo, 32, device="cuda")
    o2 = torch.empty(B, Ho, Wo, 32, device="cuda")
    ops.conv1_fwd(ops.ObsRef.from_nhwc(nhwc), w, b, o1)
    ops.conv1_fwd(ops.ObsRef.from_tensor(nchw), w, b, o2)
    ref = torch.relu(torch.nn.functional.conv2d(nchw.cpu() / 255.0, w.cpu(), b.cpu(), stride=2))
    assert rel_err(o1.permute(0, 3, 1, 2).cpu(), ref) <= RTOL
    # (the float NHWC source takes the row-walk kernel, the NCHW tensor contract the banded one: same products, summed
    #  in another order)
    assert rel_err(o1.cpu(), o2.cpu()) <= 2e-6
    g = torch.randn(B, Ho, Wo, 32, device="cuda")
    ws = torch.empty(ops.wgrad_workspace_floats(C), device="cuda")
    dw1, dw2 = torch.empty(32, C, 3, 3, device="cuda"), torch.empty(32, C, 3, 3, device="cuda")
    db1, db2 = torch.empty(32, device="cuda"), torch.empty(32, device="cuda")
    ops.conv1_wgrad(ops.ObsRef.from_nhwc(nhwc), g, dw1, db1, ws)
    ops.conv1_wgrad(ops.ObsRef.from_tensor(nchw), g, dw2, db2, ws)
    # (row-walk kernel for the NHWC source, banded kernel for the NCHW contract: the same sums in another order)
    assert rel_err(dw1.cpu(), dw2.cpu()) <= 5e-6 and rel_err(db1.cpu(), db2.cpu()) <= 5e-6
    xr = (nchw.cpu() / 255.0).requires_grad_(False)
    wt = w.cpu().clone().requires_grad_(True)
    bt = b.cpu().clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wt, bt, stride=2).backward(g.permute(0, 3, 1, 2).cpu())
    assert rel_err(dw1.cpu(), wt.grad) <= RTOL and rel_err(db1.cpu(), bt.grad) <= RTOL


@pytest.mark.parametrize("aug_name", ["color_jiggle", "noisy_cover"])
def test_update_with_float_augmentations(aug_name):
    """BASELINE configs[4] augmentation path end to end: agent.update() from the ring through the jitter /
    cover kernels and the NHWC float loader; sample_cpc() returns the reference's float NCHW tensors."""
    import curla_amd
    np.random.seed(5)
    torch.manual_seed(5)
    hw = (40, 44)
    aug = curla_amd.make_augmentor(aug_name, hw)
    agent = curla_amd.CurlSacAgent((12,) + hw, (2,), torch.device("cuda"), aug, hidden_dim=64, num_layers=4, log_interval=1)
    rb = curla_amd.ReplayBuffer((12,) + hw, (2,), 32, 6, torch.device("cuda"), aug)
    rs = np.random.RandomState(1)
    for i in range(12):
        rb.add(rs.randint(0, 256, (12,) + hw, dtype=np.uint8), rs.uniform(-1, 1, 2), rs.randn(), rs.randint(0, 256, (12,) + hw, dtype=np.uint8), False)
    obs, act, rew, nxt, nd, kw = rb.sample_cpc()
    assert obs.shape == (6, 12) + hw and obs.dtype == torch.float32
    assert float(obs.min()) >= 0 and float(obs.max()) <= 255.001
    assert not torch.equal(obs, kw["obs_pos"])  # anchor and positive are augmented independently

    class L:
        s = {}

        def log(self, k, v, step, n=1):
            self.s[k] = float(v.item() if isinstance(v, torch.Tensor) else v)
    for step in range(2):
        agent.update(rb, L(), step)
    torch.cuda.synchronize()
    for k in ("train_critic/loss", "train_actor/loss", "train/curl_loss"):
        assert np.isfinite(L.s[k]), k
