"""CPU: the index algebra of the embedder's two dedicated convolution kernels (csrc/resnet.hip: stem_conv_k, conv3x3_c32_k), emulated lane by
lane (tools/emulate_conv_layouts.py) and held against a direct convolution: fragment order of the weights, the byte / float a lane reads
for every k-pair, the tile's zero border and the masks that decide which slots are loaded, the tiles' pixel order.  (The GPU tests hold the
kernels' descriptors against the oracle; tools/bench_embed.py prints their checksum, unchanged since the generic kernel.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import emulate_conv_layouts as E  # noqa: E402


@pytest.mark.parametrize("oy0", [0, 32, 68])
def test_first_layer_from_the_bytes_of_the_chip(oy0):
    rng = np.random.default_rng(10 + oy0)
    chip = rng.integers(0, 256, (150, 150, 3), dtype=np.uint8)
    w = rng.normal(size=(32, 3, 7, 7))
    frag = E.stem_frag(E.generic_weights_rgb(w))
    assert np.count_nonzero(frag == 0.0) == 7 * 32            # the pad of each tap row's eleventh pair (half-wave 1), nothing else
    got, want = E.stem_block(chip, frag, oy0), E.stem_direct(chip, w, oy0)
    assert np.abs(got - want).max() < 1e-9


@pytest.mark.parametrize("band", range(5))
def test_band_of_the_32_channel_stage(band):
    rng = np.random.default_rng(20 + band)
    x = rng.normal(size=(35, 35, 32))
    w = rng.normal(size=(32, 32, 3, 3))
    frag = E.conv_frag(E.generic_weights_c32(w), 288)
    got, want = E.c32_band(x, frag, band), E.c32_direct(x, w, band)
    assert np.abs(got - want).max() < 1e-9
