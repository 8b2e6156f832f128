"""Boundary hardening and the two-stream schedule of an NFE (next #1a), on a real MI355X:
  * the condition cache and the decoder context carry a header (batch, layout, configuration) that every consuming C-ABI call
    checks on the host BEFORE any launch - a cache built for another batch is an error message, not an out-of-bounds read;
  * a COPY of a cache (another device pointer) is accepted after its header has been read back once;
  * rald_dit_denoise at B >= two_stream_min_batch runs two half-batches on two HIP streams: bit-identical to the unsplit launch
    sequence and to two sequential half-batch calls.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _transformer(depth=2):
    from rald_amd import models_radar_generation as G, weights
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix=""), 0), strict=True)
    return m.cuda()


def test_two_stream_nfe_is_bit_identical_to_one_stream_and_to_two_half_batches():
    from rald_amd import synth
    m = _transformer(2)
    h = m._handle(512, 64)
    B = 128
    x = synth.latents(range(B)).cuda()
    cond = synth.cond_tokens(B).cuda()
    h.set_sigmas([0.7])
    cache = h.encode_cond_tokens(cond)
    from rald_amd._lib import lib
    assert lib().rald_dit_two_stream_min_batch(h._h) == 256          # the shipped default (B = 128 gains nothing from the split)
    h.set_two_stream_min_batch(128)
    split = h.denoise(x, cache, 0)
    split2 = h.denoise(x, cache, 0)
    assert torch.equal(split, split2)                                 # run to run
    h.set_two_stream_min_batch(0)
    whole = h.denoise(x, cache, 0)
    assert torch.equal(split, whole)
    # the halves as batches of their own (their own caches: per-sample values are the same)
    for lo, hi in ((0, 64), (64, 128)):
        c = h.encode_cond_tokens(cond[lo:hi].contiguous())
        part = h.denoise(x[lo:hi].contiguous(), c, 0)
        assert torch.equal(part, split[lo:hi]), (lo, hi)
    # per-sample sigmas (training-style [B,1,1]) through the split: row b0 + i of the table for sample i of the second half
    h.set_two_stream_min_batch(128)
    sig = [0.05 + 0.01 * i for i in range(B)]
    h.set_sigmas(sig)
    a = h.denoise(x, cache, 0, per_sample=True)
    h.set_two_stream_min_batch(0)
    b = h.denoise(x, cache, 0, per_sample=True)
    assert torch.equal(a, b)
    assert not torch.equal(a[:1], a[1:2])


def test_two_stream_sampler_matches_unsplit_sampler():
    from rald_amd import synth
    m = _transformer(2)
    h = m._handle(512, 64)
    B = 130                                                           # ragged: 64 + 66
    lat = synth.latents(range(B)).cuda()
    cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
    h.set_two_stream_min_batch(128)
    s1 = h.sample(lat, cache, 4)
    h.set_two_stream_min_batch(0)
    s0 = h.sample(lat, cache, 4)
    assert torch.equal(s0, s1)


def test_condition_cache_of_another_batch_is_refused_before_any_launch():
    from rald_amd import synth
    from rald_amd._lib import lib
    m = _transformer(2)
    h = m._handle(512, 64)
    h.set_sigmas([1.0])
    c2 = h.encode_cond_tokens(synth.cond_tokens(2).cuda())
    x3 = synth.latents(range(3)).cuda()
    with pytest.raises(RuntimeError, match="does not belong to a batch of 3"):
        h.denoise(x3, c2, 0)                                          # Python layer: size check
    out = torch.empty_like(x3)
    big = torch.zeros(lib().rald_dit_cond_cache_bytes(h._h, 3), dtype=torch.uint8, device="cuda")
    big[:c2.numel()] = c2                                             # right size for batch 3, header says batch 2
    rc = lib().rald_dit_denoise(h._h, C.c_void_p(x3.data_ptr()), 3, 0, 0, C.c_void_p(big.data_ptr()), C.c_void_p(out.data_ptr()), 0,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"built for batch 2, used with batch 3" in lib().rald_last_error()
    junk = torch.zeros_like(big)
    rc = lib().rald_dit_denoise(h._h, C.c_void_p(x3.data_ptr()), 3, 0, 0, C.c_void_p(junk.data_ptr()), C.c_void_p(out.data_ptr()), 0,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"no header found" in lib().rald_last_error()
    rc = lib().rald_dit_sample(h._h, C.c_void_p(x3.data_ptr()), 3, C.c_void_p(big.data_ptr()), 4, C.c_float(0.002), C.c_float(80.0), C.c_float(7.0),
                               C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"batch 2" in lib().rald_last_error()
    # a COPY of a valid cache (new device pointer) is accepted once its header has been read back, with the same result
    x2 = synth.latents(range(2)).cuda()
    ref = h.denoise(x2, c2, 0)
    cpy = c2.clone()
    assert torch.equal(h.denoise(x2, cpy, 0), ref)
    # a cache of a handle with another configuration
    m4 = _transformer(1)
    h4 = m4._handle(512, 64)
    h4.set_sigmas([1.0])
    c_other = h4.encode_cond_tokens(synth.cond_tokens(2).cuda())
    fake = torch.zeros(lib().rald_dit_cond_cache_bytes(h._h, 2), dtype=torch.uint8, device="cuda")
    fake[:64] = c_other[:64]
    with pytest.raises(RuntimeError, match="another configuration"):
        h.denoise(x2, fake, 0)


def test_module_forward_refuses_mismatched_condition_batch():
    from rald_amd import synth
    m = _transformer(1)
    with pytest.raises(RuntimeError, match="one set of condition tokens per sample"):
        m(synth.latents(range(3)).cuda(), torch.tensor([0.1]), cond=synth.cond_tokens(2).cuda())


def test_decoder_context_of_another_batch_is_refused():
    from rald_amd import bench_ae, synth
    from rald_amd._lib import lib
    vae = bench_ae.build_ae()
    h = vae._handle()
    z2 = synth.normal([2, 512, 32], 5).cuda()
    ctx2 = h.decode_latents(z2)
    q3 = synth.queries(3, 256).cuda()
    with pytest.raises(RuntimeError, match="does not belong to a batch of 3"):
        h.decode_queries(ctx2, q3)
    out = torch.empty(3, 256, device="cuda")
    big = torch.zeros(lib().rald_ae_ctx_bytes(h._h, 3), dtype=torch.uint8, device="cuda")
    big[:ctx2.numel()] = ctx2
    rc = lib().rald_ae_decode_queries(h._h, C.c_void_p(big.data_ptr()), C.c_void_p(q3.data_ptr()), 3, 256, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"built for batch 2, used with batch 3" in lib().rald_last_error()
    q2 = synth.queries(2, 256).cuda()
    ref = h.decode_queries(ctx2, q2)
    assert torch.equal(h.decode_queries(ctx2.clone(), q2), ref)


def test_results_do_not_depend_on_what_the_lds_held_before():
    """VERDICT r02 weak #4 / ADVICE: a dropped kernel variant gave different results only when other streams shared the chip.  LDS
    is not cleared between workgroups: a read of a never-written LDS word is invisible back to back and wrong under concurrency.
    Every product path must give bit-identical (and finite) results after all LDS of the chip has been filled with NaN patterns."""
    from rald_amd import bench_ae, config, models_radar_generation as G, synth, weights
    from rald_amd._lib import check, lib
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    m = G.EDMPrecond(n_latents=512, channels=32, depth=2, configs=config.shipped_generation_config())
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=2), 0), strict=True)
    m = m.cuda()
    vae = bench_ae.build_ae()
    h = m._handle()
    cases = {}
    for B in (1, 2, 8, 64):                                          # small-batch fused kernels, mid engines, the large-batch engines
        x = synth.latents(range(B)).cuda()
        cond = synth.cond_tokens(B, seed=5).cuda()

        def nfe(x=x, cond=cond):
            h.set_sigmas([1.3])
            return h.denoise(x, h.encode_cond_tokens(cond), 0)
        cases[f"nfe_B{B}"] = nfe
    cube = synth.radar_cube(1).cuda()
    cases["cond_encode"] = lambda: h.encode_cond(cube)[0]
    cases["sample_B1"] = lambda: h.sample(synth.latents([3]).cuda(), h.encode_cond(cube)[1], 3, use_graph=False)
    pc, eps, q = synth.structured_cloud(1, 10000).cuda(), synth.normal([1, 512, 32], 3), synth.structured_queries(1, 5000).cuda()
    ah = vae._handle()
    cases["ae_encode"] = lambda: ah.encode(pc, eps)[1]
    z = synth.normal([2, 512, 32], 9).cuda()
    cases["ae_decode"] = lambda: ah.decode_queries(ah.decode_latents(z[:1].contiguous(), use_graph=False), q)
    cases["ae_decode_B2"] = lambda: ah.decode_queries(ah.decode_latents(z, use_graph=False), torch.cat([q, q]))
    for name, fn in cases.items():
        ref = fn()
        torch.cuda.synchronize()
        check(lib().rald_debug_poison_lds(st()))
        out = fn()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), name
        assert torch.equal(out, ref), name


def test_two_stream_autotune_keeps_results_and_reports_its_choice():
    """Opt-in per-box choice between the whole-batch and the two-half-batch schedule of an NFE (128-255 samples): whatever it picks, the
    result is the bits of the untuned call, the choice is recorded once per batch size, and the library's threshold follows it."""
    from rald_amd import synth
    from rald_amd._lib import lib
    m = _transformer(2)
    h = m._handle(512, 64)
    B = 128
    x, cache = synth.latents(range(B)).cuda(), h.encode_cond_tokens(synth.cond_tokens(B).cuda())
    h.set_sigmas([0.9])
    ref = h.denoise(x, cache, 0)
    assert not h._two_stream_tuned and lib().rald_dit_two_stream_min_batch(h._h) == 256
    h.autotune_two_stream = True
    out = h.denoise(x, cache, 0)
    assert torch.equal(out, ref)
    split, ms_whole, ms_split = h._two_stream_tuned[B]
    assert ms_whole > 0 and ms_split > 0
    assert lib().rald_dit_two_stream_min_batch(h._h) == (B if split else 256)
    assert torch.equal(h.denoise(x, cache, 0), ref) and len(h._two_stream_tuned) == 1      # tuned once
