"""GPU: the small dense HIP primitives (encoder tails, trunk) against plain torch autograd on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from spvipes_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


def _torch_tail(enc, h, eps, training):
    import torch.nn.functional as F
    h = F.relu(enc.fc2(h))
    loc, logvar = enc.mu_encoder(h), enc.lvar_encoder(h)
    scale = (0.5 * logvar).exp()
    logz = loc + scale * eps
    kl = (0.5 * (scale * scale + loc * loc - 1 - logvar)).sum(1)
    return loc, logvar, scale, logz, F.softmax(logz, -1), kl


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,H,ns", [(100, 24, (3, 6, 5, 9)), (700, 128, (10, 25, 10, 25))])
def test_encoder_tails_match_torch(dev, training, B, H, ns):
    import copy
    from spvipes_amd.module import Encoder
    from spvipes_amd.nn_ops import EncoderSpec, EncoderTails
    from spvipes_amd.ops import Workspace
    torch.manual_seed(0)
    encs = [Encoder(40, n, H, 0.0).to(dev) for n in ns]
    for e in encs:
        for bn in (e.mu_encoder[1], e.lvar_encoder[1]):
            bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
        e.train(training)
    refs = copy.deepcopy(encs)
    h1 = [torch.randn(B, 2 * H, device=dev).requires_grad_(True) for _ in range(2)]
    h1r = [t.detach().clone().requires_grad_(True) for t in h1]
    eps = [torch.randn(B, n, device=dev) for n in ns]
    specs = [EncoderSpec(encs[0], 0, 0), EncoderSpec(encs[1], 0, H), EncoderSpec(encs[2], 1, 0), EncoderSpec(encs[3], 1, H)]
    flat = [p for s in specs for p in s.params()]
    outs = EncoderTails.apply(specs, eps, training, 0.0, 0, Workspace(dev), None, *h1, *flat)
    want = []
    for i, (e, s) in enumerate(zip(refs, specs)):
        want += list(_torch_tail(e, h1r[s.h1_group][:, s.h1_col:s.h1_col + H], eps[i], training))
    gen = torch.Generator(device=dev).manual_seed(1)
    coef = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
    use = lambda k: (k % 6) != 4  # theta is not differentiable through the fused op
    sum((o * c).sum() for k, (o, c) in enumerate(zip(outs, coef)) if use(k)).backward()
    sum((o * c).sum() for k, (o, c) in enumerate(zip(want, coef)) if use(k)).backward()
    for k, (o, w) in enumerate(zip(outs, want)):
        torch.testing.assert_close(o, w, rtol=2e-4, atol=2e-5, msg=lambda m: f"output {k}: {m}")
    for a, b in zip(h1, h1r):
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-3, atol=2e-4 * float(b.grad.abs().max()))
    for e, r in zip(encs, refs):
        gmax = max(float(q.grad.abs().max()) for q in r.parameters() if q.grad is not None)  # biases ahead of a train-mode BN have zero gradient: noise floor
        for (name, p), (_, q) in zip(e.named_parameters(), r.named_parameters()):
            if q.grad is None and p.grad is None:
                continue  # fc1 is not part of the tail
            g = torch.zeros_like(q) if q.grad is None else q.grad
            torch.testing.assert_close(p.grad, g, rtol=2e-3, atol=3e-4 * max(float(g.abs().max()), 0.05 * gmax), msg=lambda m: f"{name}: {m}")
        if training:
            for bn, rbn in ((e.mu_encoder[1], r.mu_encoder[1]), (e.lvar_encoder[1], r.lvar_encoder[1])):
                torch.testing.assert_close(bn.running_mean, rbn.running_mean, rtol=1e-4, atol=1e-6)
                torch.testing.assert_close(bn.running_var, rbn.running_var, rtol=1e-4, atol=1e-6)


def test_dropout_mask_is_reproducible_and_has_the_right_rate(dev):
    from spvipes_amd.module import Encoder
    from spvipes_amd.nn_ops import EncoderSpec, EncoderTails
    from spvipes_amd.ops import Workspace
    torch.manual_seed(0)
    B, H = 2048, 64
    enc = Encoder(16, 5, H, 0.25).to(dev).train()
    h1 = torch.randn(B, 2 * H, device=dev).requires_grad_(True)
    eps = [torch.randn(B, 5, device=dev)]
    spec = [EncoderSpec(enc, 0, 0)]
    run = lambda seed: EncoderTails.apply(spec, eps, True, 0.25, seed, Workspace(dev), None, h1, *spec[0].params())
    a, b, c = run(7), run(7), run(8)
    assert torch.equal(a[3], b[3]) and not torch.equal(a[3], c[3])
    # drop rate: compare kept fraction of the positive fc2 activations through a backward pass
    a[3].sum().backward()
    assert torch.isfinite(h1.grad).all()


def test_spv_randn_is_standard_normal_and_keyed_by_the_device_counter():
    """spv_randn (Philox 4x32-10 + Box-Muller): moments of a large draw, independence of draws under different counters / keys,
    determinism for the same (key, counter), odd lengths and unaligned outputs."""
    import torch
    from spvipes_amd import _abi
    dev = torch.device("cuda:0")
    n = 1 << 22
    ctr = torch.zeros((), dtype=torch.int64, device=dev)
    def draw(n_, key, out=None):
        out = torch.empty(n_, dtype=torch.float32, device=dev) if out is None else out
        _abi.call("spv_randn", _abi.ptr(out), n_, _abi.ptr(ctr), key, _abi.stream_ptr())
        return out
    a = draw(n, 1234)
    assert abs(float(a.mean())) < 3e-3 and abs(float(a.std()) - 1.0) < 3e-3
    assert abs(float((a ** 3).mean())) < 1e-2 and abs(float((a ** 4).mean()) - 3.0) < 3e-2
    assert float(a.abs().max()) < 7.0 and bool(torch.isfinite(a).all())
    # fraction inside one sigma
    assert abs(float((a.abs() < 1).float().mean()) - 0.682689) < 2e-3
    b = draw(n, 1234)
    assert torch.equal(a, b)                      # same key, same counter
    ctr.add_(1)
    c = draw(n, 1234)
    d = draw(n, 1235)
    for x, y in ((a, c), (a, d), (c, d)):
        assert abs(float((x * y).mean())) < 3e-3  # uncorrelated streams
    # neighbouring elements uncorrelated, odd length, unaligned start
    assert abs(float((a[:-1] * a[1:]).mean())) < 3e-3
    buf = torch.full((1003,), 7.0, device=dev)
    draw(1001, 99, out=buf[1:1002])
    assert float(buf[0]) == 7.0 and float(buf[1002]) == 7.0 and bool((buf[1:1002] != 7.0).all())
    assert torch.equal(buf[1:1002], draw(1001, 99))


@pytest.mark.parametrize("B0,B1,nlab", [(64, 64, 3), (1000, 777, 10), (4096, 4096, 10), (5000, 9000, 37), (4096, 300, 1)])
def test_label_pairing_matches_the_reference_rule(B0, B1, nlab):
    """spv_poe_partner (rank within label + partner lookup) against a plain Python restatement of the reference's pairing rule
    (module/spVIPESmodule.py:685-701): cell i with label L and rank k among the same-label cells of its minibatch (batch order) gets the
    k-th cell of label L of the other minibatch (mode 0), the padding expert if L occurs there fewer than k + 1 times (mode 1), the
    dummy expert if L does not occur there (mode 2).  Includes minibatches beyond the 4096 cells whose chunks stay in registers."""
    import numpy as np
    import torch
    from spvipes_amd import ops
    from spvipes_amd.nn_ops import label_partners
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B0 + B1 + nlab)
    labs = [rng.integers(0, nlab, size=B0), rng.integers(0, nlab + 2, size=B1)]   # group 1 also has labels group 0 never sees
    t = [torch.tensor(l, dtype=torch.float32, device=dev).unsqueeze(1) for l in labs]
    # (a) the stand-alone pairing entry point (rank + lookup kernels)
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, stream_ptr
    lab = [x.flatten().contiguous() for x in t]
    i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)
    order, rank, partner_a, mode_a = [i32(B0), i32(B1)], [i32(B0), i32(B1)], [i32(B0), i32(B1)], [i32(B0), i32(B1)]
    tables, err = torch.empty(2, 2, 1024, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    _abi.call("spv_poe_partner", ptr(lab[0]), ptr(lab[1]), B0, B1, ptr(order[0]), ptr(order[1]), ptr(rank[0]), ptr(rank[1]), ptr(tables),
              ptr(partner_a[0]), ptr(mode_a[0]), ptr(partner_a[1]), ptr(mode_a[1]), ptr(err), stream_ptr())
    # (b) the shipped path: spv_poe_rank, then the lookup INSIDE the fusion kernel (PoELabel), which stores partner / mode for its backward
    from spvipes_amd.nn_ops import PoELabel
    n = 5
    gen = torch.Generator(device=dev).manual_seed(1)
    mk = lambda B: torch.randn(B, n, generator=gen, device=dev)
    ws = ops.Workspace(dev)
    pre = label_partners(t, ws)
    PoELabel.apply([t[0], t[1]], [mk(B0), mk(B1)], ws, pre, mk(B0), mk(B0), mk(B1), mk(B1))
    partner, mode = pre[0], pre[1]
    torch.cuda.synchronize()
    assert int(err) == 0
    for g in (0, 1):
        assert torch.equal(partner[g], partner_a[g]) and torch.equal(mode[g], mode_a[g])
    for g in (0, 1):
        o = 1 - g
        where = {}
        for j, L in enumerate(labs[o]):
            where.setdefault(int(L), []).append(j)
        seen, want_p, want_m = {}, [], []
        for L in labs[g]:
            k = seen.get(int(L), 0)
            seen[int(L)] = k + 1
            cells = where.get(int(L), [])
            if k < len(cells):
                want_p.append(cells[k]); want_m.append(0)
            else:
                want_p.append(-1); want_m.append(1 if cells else 2)
        assert mode[g].cpu().tolist() == want_m
        assert partner[g].cpu().tolist() == want_p
