#!/usr/bin/env python3
"""Host-side fuzz of the library's shape queries and argument checks, meant to run against a build whose HOST code is
instrumented with AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py builds it; the GPU code is the
shipped code, the sanitizers cannot run on the GPU here).  No GPU is needed: only entry points that answer from host arithmetic
are called with arbitrary shapes (tile geometry, split-K / split planning, workspace sizes, 2 GiB image-run chunking, staging
rules), and launch entry points only with arguments they must REJECT before touching anything.

r04 findings, fixed: ad_conv3x3_wgrad_ws_bytes divided by zero for cin below the channel granule (a query with cin = 3);
ad_conv3x3_fwd_ws_bytes / plan_wgrad / pw_wgrad_plan overflowed 32-bit tile / pixel counts on shapes beyond 2^31 pixels.
"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adunet_amd import _lib
lib = _lib.load()
rnd = random.Random(1)
for _ in range(3000):
    nn = rnd.choice([1,2,3,7,8,16,64,70,1024]); h = rnd.choice([1,2,3,4,7,16,17,34,56,93,154,256,512,1000]); w = rnd.choice([1,2,4,16,33,34,56,180,256,512])
    cin = rnd.choice([3,16,32,64,96,128,256,512,1024,2048,4096]); cout = rnd.choice([16,32,64,96,128,256,512,1024,4096]); dt = rnd.choice([0,1,2,5])
    lib.ad_conv3x3_fwd_ws_bytes(nn,h,w,cin,cout,dt); lib.ad_conv3x3_wgrad_ws_bytes(nn,h,w,cin,cout,dt)
    lib.ad_conv3x3_ln_relu_is_fused(nn,h,w,cin,0,cout,dt); lib.ad_conv3x3_dgrad_relu_is_fused(nn,h,w,cin,cout,cout//2,dt)
    lib.ad_conv3x3_dgrad_ln_bwd_is_fused(nn,h,w,cin,cout,dt); lib.ad_conv3x3_c3_supported(nn,h,w,cout,dt)
    lib.ad_conv3x3_mosaic(nn,h,w,cin,0,cout,dt,0); lib.ad_conv3x3_mosaic(nn,h,w,cin//2,cin//2,cout,dt,1)
    lib.ad_pw_supported(nn*h*w,cin,9*cout,dt); lib.ad_pw_gemm_variant(nn*h*w,cin,9*cout,dt); lib.ad_pw_wgrad_supported(nn*h*w,cin,cout,dt); lib.ad_pw_wgrad_ws_bytes(nn*h*w,cin,cout)
    lib.ad_resample_ln_bwd_ws_bytes(nn,h,w,cout,dt) if hasattr(lib,'ad_resample_ln_bwd_ws_bytes') else None
    lib.ad_layernorm_bwd_ws_bytes(nn*h*w,cout); lib.ad_head_ws_bytes(nn,cout); lib.ad_head_ln_bwd_ws_bytes(nn,cout); lib.ad_conv3x3_pack_elems(cin,cout,max(cin,16)); lib.ad_metrics_ws_bytes(nn,h,w)
rnd = random.Random(7)
for _ in range(4000):
    nn = rnd.choice([1,3,64,512,4096,65536]); h = rnd.choice([1,5,64,256,1024,4096,30000]); w = rnd.choice([1,7,64,256,2048,30000])
    cin = rnd.choice([16,32,64,128,1024,4096,16384]); cout = rnd.choice([16,64,128,1024,4096,16384]); dt = rnd.choice([0,1,2])
    lib.ad_conv3x3_fwd_ws_bytes(nn,h,w,cin,cout,dt); lib.ad_conv3x3_wgrad_ws_bytes(nn,h,w,cin,cout,dt)
    lib.ad_conv3x3_ln_relu_is_fused(nn,h,w,cin,0,cout,dt); lib.ad_conv3x3_dgrad_relu_is_fused(nn,h,w,cin,cout,max(cout//2,16),dt)
    lib.ad_conv3x3_dgrad_ln_bwd_is_fused(nn,h,w,cin,cout,dt); lib.ad_conv3x3_c3_supported(nn,h,w,cout,dt)
    m = nn*h*w
    lib.ad_conv3x3_mosaic(nn,h,w,cin,0,cout,dt,0); lib.ad_conv3x3_mosaic(nn,h,w,cin,0,cout,dt,1)
    lib.ad_pw_supported(m,cin,9*cout,dt); lib.ad_pw_gemm_variant(m,cin,9*cout,dt); lib.ad_pw_wgrad_supported(m,cin,cout,dt); lib.ad_pw_wgrad_ws_bytes(m,cin,cout)
    lib.ad_resample_ln_bwd_ws_bytes(nn,h,w,cout,dt); lib.ad_resample_ln_bwd_supported(nn,h,w,cout,rnd.choice([1,2,8,40]),dt)
    lib.ad_layernorm_bwd_ws_bytes(m,cout); lib.ad_head_ws_bytes(nn,cout); lib.ad_head_ln_bwd_ws_bytes(nn,cout); lib.ad_metrics_ws_bytes(nn,h,w)
    lib.ad_conv3x3_c3_wgrad_ws_bytes(nn,h,w); lib.ad_conv3x3_pack_elems(cin,cout,1); lib.ad_conv3x3_pack_job_blocks(cin,cout)
    lib.ad_upconv_gather_fwd_supported(cout, rnd.choice([1,7,100,100000]), dt); lib.ad_upconv_gather_bwd_supported(rnd.choice([0,1,8,14,15,1000]))
    ow = rnd.choice([1,2,16,100,1000]); ww = rnd.choice([1,3,50,999])
    sx = np.sort(np.array([rnd.randrange(0, ww) for _ in range(ow)], dtype=np.int32))
    lib.ad_upconv_slab_cols(sx.ctypes.data, ww, ow, cout, dt)
# launch entry points must reject bad arguments before touching anything
assert lib.ad_conv3x3_fwd(None, 64, None, 0, None, None, None, 64, None, 1, 16, 16, 64, 0, None, 0, 1, None) != 0
assert lib.ad_conv3x3_wgrad(None, 64, None, 0, None, None, 64, 1, 16, 16, 64, None, 0, 1, None) != 0
assert lib.ad_layernorm_relu_bwd(None, None, None, None, None, None, None, None, None, None, 0, 64, 1, None, 0, 1, None) != 0
assert lib.ad_pw_gemm(None, None, None, 4096, 100, 576, 1, None) != 0
print("ok")
