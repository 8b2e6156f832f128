import sys, torch; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_boundary import _transformer
from rald_amd import synth
with torch.no_grad():
    m = _transformer(2); h = m._handle(512, 64); B = 128
    x = synth.latents(range(B)).cuda(); cond = synth.cond_tokens(B).cuda(); h.set_sigmas([0.7])
    cache = h.encode_cond_tokens(cond)
    h.set_two_stream_min_batch(0)
    whole = h.denoise(x, cache, 0)
    c0 = h.encode_cond_tokens(cond[:64].contiguous())
    part = h.denoise(x[:64].contiguous(), c0, 0)
    print("whole[:64] == part", torch.equal(whole[:64], part), float((whole[:64]-part).abs().max()))
    # compare cache pieces: K rows of the first 64 samples
    T, L, D = 64, 2, 512
    k128 = cache[64:64 + 128*T*L*D*2].view(torch.bfloat16).view(128*T, L*D)
    k64 = c0[64:64 + 64*T*L*D*2].view(torch.bfloat16).view(64*T, L*D)
    print("K equal", torch.equal(k128[:64*T], k64))
    part2 = h.denoise(x[:64].contiguous(), c0, 0)
    print("repeat equal", torch.equal(part, part2))
