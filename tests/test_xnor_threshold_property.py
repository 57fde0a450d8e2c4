"""The argument behind conv_xnor.hip's sign-only epilogue, checked in float32 on the CPU (numpy's float32 multiply and
add round like v_mul_f32 / v_add_f32): for mean >= 0 the result fl(fl((2*count - K) * mean) + bias) of the reference's
XNOR convolution (gemm_nn_custom_bin_mean_transposed, src/additionally.c:1531: `(2*count - K) * mean_val`, then
`+ bias`) is a non-decreasing function of the integer match count, so (result > 0) is a step at one threshold per
filter.  xnor_threshold_kernel finds that threshold by evaluating every count on the device and refuses filters that
are not a step; the GPU test tests/test_gpu_int8_xnor.py::test_xnor_sign_thresholds_match_the_float_epilogue compares
the device's thresholds with this same numpy expression.  No product code runs here."""
import numpy as np
import pytest


def _positive(K, mean, bias):
    c = np.arange(K + 1, dtype=np.int64)
    with np.errstate(over="ignore", invalid="ignore"):          # +-inf products are part of the case list
        v = ((2 * c - K).astype(np.float32)[None, :] * mean[:, None]).astype(np.float32) + bias[:, None]
        return v.astype(np.float32) > 0


@pytest.mark.parametrize("K", [9, 144, 576, 4608, 9216])
def test_sign_is_a_step_function_of_the_count_for_nonnegative_mean(K):
    rng = np.random.default_rng(K)
    n = 4096
    mean = np.abs(rng.normal(0, 0.05, n)).astype(np.float32)
    mean[:64] = 0.0                                              # constant result: threshold 0 or K + 1
    mean[64:128] = np.float32(1e-38)                             # products in the subnormal range
    mean[128:192] = np.float32(3e38) / np.float32(K)             # products near overflow
    bias = rng.normal(0, 1.0, n).astype(np.float32)
    bias[::7] = 0.0
    with np.errstate(over="ignore"):
        bias[1::7] = -bias[1::7] * np.float32(K) * mean[1::7]    # thresholds spread over the whole range
    pos = _positive(K, mean, bias)
    first = np.where(pos.any(axis=1), pos.argmax(axis=1), K + 1)
    c = np.arange(K + 1)
    assert np.array_equal(pos, c[None, :] >= first[:, None])
    assert (first == 0).any() and (first == K + 1).any() and ((first > 0) & (first <= K)).any()


def test_negative_mean_is_not_a_step():
    """why the device kernel verifies instead of assuming: a (hypothetical) negative mean flips the direction"""
    pos = _positive(144, np.array([-0.1], np.float32), np.array([0.5], np.float32))[0]
    first = int(pos.argmax())
    assert first == 0 and not pos[-1]                            # positive at count 0, not positive at count K
