"""bench.py's byte model: the algorithmic bytes per stage the roofline fractions are computed from, and the bytes this
build's layout moves (triangular coefficients, folded table) that the line reports beside them."""
import bench


def test_stage_model_bytes_at_the_headline_shape():
    m = bench.stage_model()
    C, H, W, L, M = 384, 180, 360, 180, 181
    act, coef = C * H * W * 4, C * L * M * 8
    # forward SHT = field + Legendre table + coefficients, each once: 223.1 MB
    sht = 384 * 180 * 360 * 4 + 181 * 180 * 180 * 4 + 384 * 180 * 181 * 8
    assert abs(sht / 1e6 - 223.1) < 0.1
    assert m["forward_transform.dft"][1] == act + coef
    assert m["mlp.fc2+outer_skip"][1] == 2 * act + 768 * H * W * 4
    assert abs(m["dhconv"][1] / 1e6 - 412.5) < 0.1                      # coefficients in + filter (212.3 MB) + coefficients out
    for k, (fl, alg, moved) in m.items():
        assert 0 < moved <= alg, k                                        # the layout never moves more than the dense operands
    tri = C * 8 * sum(min(l + 1, M) for l in range(L))
    assert m["forward_transform.legendre"][2] == coef + 8_600_000 + tri  # X + folded table + the l >= m triangle
    assert m["dhconv"][2] == 2 * tri + 2 * C * C * L * 4
    s80 = bench.stage_model(cout=72)
    assert s80["decoder"][1] - m["decoder"][1] == 22 * H * W * 4
