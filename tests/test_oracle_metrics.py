"""Known answers that pin the FID / IS / KID restatement (oracle/inception_ref.py; torch-fidelity 0.3.0's defaults as the reference calls
them, utils_training.py:948-1001, utils_Img2Img.py:462-563) and the product's host-side statistics (phendiff_amd/metrics.py) against it."""
import numpy as np
import pytest
import torch

from oracle import (InceptionV3FeaturesRef, fid_from_statistics_ref, fid_statistics_ref, isc_ref, kid_ref, randomize_inception_,
                    tf1_bilinear_resize_ref)


def test_inception_structure_matches_the_public_parameter_count():
    """torchvision's inception_v3 without its auxiliary head has 23 834 568 parameters (27 161 264 with it, 3 326 696 in the head); the FID
    network is that structure with a 1008-way fc: + 8 x 2049."""
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(InceptionV3FeaturesRef(1000)) == 23_834_568
    m = InceptionV3FeaturesRef()
    assert n(m) == 23_834_568 + 8 * 2049
    names = set(m.state_dict())
    for k in ("Conv2d_1a_3x3.conv.weight", "Conv2d_4a_3x3.bn.running_var", "Mixed_5b.branch5x5_2.conv.weight", "Mixed_6a.branch3x3dbl_3.bn.bias",
              "Mixed_6e.branch7x7dbl_5.conv.weight", "Mixed_7a.branch7x7x3_4.conv.weight", "Mixed_7c.branch3x3dbl_3b.conv.weight", "fc.weight"):
        assert k in names, k
    assert tuple(m.Mixed_6b.branch7x7_2.conv.weight.shape) == (128, 128, 1, 7) and tuple(m.Mixed_7b.branch3x3_2b.conv.weight.shape) == (384, 384, 3, 1)
    out = randomize_inception_(m)(torch.randint(0, 256, (2, 3, 40, 56), dtype=torch.uint8))
    assert tuple(out["2048"].shape) == (2, 2048) and tuple(out["logits"].shape) == (2, 1008)
    assert torch.allclose(out["logits"] - out["logits_unbiased"], m.fc.bias.expand(2, -1), atol=1e-5)


def test_tf1_resize_has_no_half_pixel_centres():
    x = torch.arange(16, dtype=torch.float32).view(1, 1, 4, 4)
    assert torch.equal(tf1_bilinear_resize_ref(x, (4, 4)), x)                      # same size: identity
    up = tf1_bilinear_resize_ref(x, (8, 8))                                        # scale 1/2: even outputs ARE the inputs, odd ones the midpoints
    assert torch.equal(up[0, 0, ::2, ::2], x[0, 0])
    assert torch.allclose(up[0, 0, 0, 1::2], torch.tensor([0.5, 1.5, 2.5, 3.0]))   # the last column clamps its right neighbour
    down = tf1_bilinear_resize_ref(x, (2, 2))                                      # scale 2: top-left samples, no averaging
    assert torch.equal(down[0, 0], torch.tensor([[0.0, 2.0], [8.0, 10.0]]))


def test_fid_is_kid_known_answers():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(400, 16))
    mu, sig = fid_statistics_ref(a)
    assert abs(fid_from_statistics_ref(mu, sig, mu, sig)) < 1e-8                   # identical sets
    # commuting (diagonal) covariances: d^2 = |mu1 - mu2|^2 + sum (sqrt(s1) - sqrt(s2))^2
    s1, s2 = rng.uniform(0.5, 2.0, 16), rng.uniform(0.5, 2.0, 16)
    m1, m2 = rng.normal(size=16), rng.normal(size=16)
    want = ((m1 - m2) ** 2).sum() + ((np.sqrt(s1) - np.sqrt(s2)) ** 2).sum()
    assert abs(fid_from_statistics_ref(m1, np.diag(s1), m2, np.diag(s2)) - want) < 1e-9
    # IS: constant logits -> p(y|x) = p(y): score 1; one-hot-ish logits over K balanced classes -> K
    assert abs(isc_ref(np.zeros((50, 7)))["inception_score_mean"] - 1.0) < 1e-12
    lg = np.full((70, 7), -50.0)
    lg[np.arange(70), np.arange(70) % 7] = 50.0
    r = isc_ref(lg, splits=1, shuffle=False)
    assert abs(r["inception_score_mean"] - 7.0) < 1e-9 and r["inception_score_std"] == 0.0
    # KID: two samples of the same distribution: ~0 (unbiased estimator, either sign); shifted: clearly positive
    b = rng.normal(size=(400, 16))
    same = kid_ref(a, b, kid_subsets=20, kid_subset_size=100)
    far = kid_ref(a, b + 1.0, kid_subsets=20, kid_subset_size=100)
    assert abs(same["kernel_inception_distance_mean"]) < 0.05 < far["kernel_inception_distance_mean"]
    with pytest.raises(AssertionError):
        kid_ref(a, b, kid_subset_size=401)


def test_product_statistics_equal_the_oracle():
    """phendiff_amd.metrics' fp64 host functions (what the product computes from the HIP features) on the same features."""
    import phendiff_amd.metrics as M
    rng = np.random.default_rng(1)
    f1, f2 = rng.normal(size=(300, 24)) * rng.uniform(0.5, 2, 24), rng.normal(size=(260, 24)) + 0.3
    lg = rng.normal(size=(300, 40)) * 3
    assert np.allclose(M.fid_from_statistics(*M.fid_statistics(f1), *M.fid_statistics(f2)),
                       fid_from_statistics_ref(*fid_statistics_ref(f1), *fid_statistics_ref(f2)), rtol=1e-12)
    a, b = M.inception_score(lg), isc_ref(lg)
    assert all(abs(a[k] - b[k]) < 1e-10 * max(1, abs(b[k])) for k in b)
    a, b = M.kernel_inception_distance(f1, f2, kid_subsets=10, kid_subset_size=50), kid_ref(f1, f2, kid_subsets=10, kid_subset_size=50)
    assert all(abs(a[k] - b[k]) < 1e-12 + 1e-10 * abs(b[k]) for k in b)
    with pytest.raises(ValueError):
        M.kernel_inception_distance(f1, f2, kid_subset_size=261)
    x = rng.random((3, 8, 8, 3)).astype(np.float32)
    assert M.to_uint8(x).dtype == np.uint8 and np.array_equal(M.to_uint8(x), (x * 255).round().astype("uint8"))
    # the module tree carries torch-fidelity's names (its pt_inception weights load) and shapes
    net, ref = M.InceptionV3Features("f32"), InceptionV3FeaturesRef()
    sa, sb = net.state_dict(), ref.state_dict()
    assert set(sa) == set(sb) and all(sa[k].shape == sb[k].shape for k in sa)
