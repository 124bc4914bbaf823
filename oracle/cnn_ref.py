"""TEST INFRASTRUCTURE ONLY: plain numpy ResNet for CIFAR-10 in the ordinary [c][h][w] layout, the function the
encrypted network of phantom-fhe_amd/cnn.py must compute (the network of the reference's examples/cnn.cu:291-378:
conv 3x3 / batch norm / ReLU, residual blocks with the parameter-free shortcut, average pooling, linear layer).
Parity unpinned by reference outputs: the reference's result/*.txt logs were produced from CIFAR-10 test images
(testFile/test_values.txt) that the repository does not hold.  Only tests/ and tools/ benches' checks import this."""
import numpy as np


def conv3x3(x, w, stride):
    """x [ci][h][w], w [co][ci][3][3], zero padding 1"""
    ci, h, wd = x.shape
    xp = np.zeros((ci, h + 2, wd + 2))
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], h, wd))
    for i1 in range(3):
        for i2 in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, i1, i2], xp[:, i1:i1 + h, i2:i2 + wd])
    return out[:, ::stride, ::stride]


def batch_norm(x, bias, mean, var, weight, eps=1e-5):
    return (x - mean[:, None, None]) * (weight / np.sqrt(var + eps))[:, None, None] + bias[:, None, None]


def shortcut_a(x):
    """every other pixel, c/2 zero channels on both sides"""
    c = x.shape[0]
    out = np.zeros((2 * c, x.shape[1] // 2, x.shape[2] // 2))
    out[c // 2:c // 2 + c] = x[:, ::2, ::2]
    return out


def resnet_cifar10(params, image, end_num, widths=(16, 32, 64), relu=None, trace=None):
    """relu: callable on arrays (default: exact); returns the logits"""
    relu = relu or (lambda v: np.maximum(v, 0.0))

    def conv_bn(x, conv, bn, co, st):
        w = params[conv + "_weight"].reshape(co, x.shape[0], 3, 3)
        return batch_norm(conv3x3(x, w, st), params[bn + "_bias"], params[bn + "_running_mean"], params[bn + "_running_var"],
                          params[bn + "_weight"])
    x = relu(conv_bn(np.asarray(image, dtype=np.float64), "conv1", "bn1", widths[0], 1))
    if trace is not None:
        trace("layer0", x)
    for j, co in enumerate(widths):
        for k in range(end_num + 1):
            stem = "layer%d_%d" % (j + 1, k)
            st = 2 if j >= 1 and k == 0 else 1
            y = relu(conv_bn(x, stem + "_conv1", stem + "_bn1", co, st))
            y = conv_bn(y, stem + "_conv2", stem + "_bn2", co, 1)
            x = relu(y + (shortcut_a(x) if st == 2 else x))
            if trace is not None:
                trace(stem, x)
    pooled = x.mean(axis=(1, 2))
    q = params["linear_bias"].size
    return params["linear_weight"].reshape(q, widths[-1]) @ pooled + params["linear_bias"]


def synthetic_parameters(end_num, widths=(16, 32, 64), in_ch=3, classes=10, seed=0):
    """seeded He-initialised weights with batch-norm statistics near (0, 1): keeps activations of order one"""
    rng = np.random.default_rng(seed)
    p = {}

    def bn(stem, c):
        p[stem + "_bias"] = rng.normal(0.0, 0.1, c)
        p[stem + "_running_mean"] = rng.normal(0.0, 0.1, c)
        p[stem + "_running_var"] = rng.uniform(0.8, 1.2, c)
        p[stem + "_weight"] = rng.uniform(0.5, 1.0, c)
    p["conv1_weight"] = rng.normal(0.0, np.sqrt(2.0 / (9 * in_ch)), widths[0] * in_ch * 9)
    bn("bn1", widths[0])
    ci = widths[0]
    for j, co in enumerate(widths):
        for k in range(end_num + 1):
            stem = "layer%d_%d" % (j + 1, k)
            p[stem + "_conv1_weight"] = rng.normal(0.0, np.sqrt(2.0 / (9 * ci)), co * ci * 9)
            bn(stem + "_bn1", co)
            p[stem + "_conv2_weight"] = rng.normal(0.0, np.sqrt(1.0 / (9 * co)), co * co * 9)
            bn(stem + "_bn2", co)
            ci = co
    p["linear_weight"] = rng.normal(0.0, 0.3, classes * widths[-1])
    p["linear_bias"] = rng.normal(0.0, 0.1, classes)
    return p
