"""scripts/kernel_names.py: the three spellings rocprofv3 gives one kernel instance map to the name the C side reports through
lg_last_kernel() (strings as they appear in profiles/r2_c3_kernel_stats.md)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from kernel_names import short  # noqa: E402

CASES = [
    ("conv_halo_kernel<bool _Accum, int, E, 4, false, true, false, 1, 4, 4, 1, true>", "conv_halo_kernel<bf16,UP,K-sliced>"),
    ("_ZN12_GLOBAL__N_116conv_halo_kernelIDF16bLi0ELi2ELb0ELb1ELb0ELi1ELi4ELi4ELi1ELb1EEEvNS_10HaloParamsE", "conv_halo_kernel<bf16,DOWN>"),
    ("_ZN12_GLOBAL__N_116conv_halo_kernelIDF16bLi1ELi4ELb0ELb1ELb1ELi1ELi4ELi4ELi1ELb0EEEvNS_10HaloParamsE", "conv_halo_kernel<UP,resident>"),
    ("_ZN12_GLOBAL__N_116conv_halo_kernelIfLi1ELi4ELb0ELb0ELb1ELi4ELi1ELi1ELi1ELb0EEEvNS_10HaloParamsE", "conv_halo_kernel<f32,UP,resident>"),
    ("void (anonymous namespace)::conv_halo_kernel<float, 1, 2, false, false, false, 2, 2, 2, 1, false>(HaloParams)", "conv_halo_kernel<f32,UP,K-sliced>"),
    ("conv_down3_kernel<true, false, false, 128>", "conv_down3_kernel<NW=128>"),
    ("conv_down3_kernel<false, true, false, 64>", "conv_down3_kernel<NW=64>"),
    ("conv_down3_kernel<true, false, true, 128>", "conv_down3_kernel<PAIR>"),
    ("conv_down3_kernel<true, false, false, 128, true>", "conv_down3_kernel<NW=128,NORM>"),
    ("conv_down3_kernel<true, false, false, 128, false>", "conv_down3_kernel<NW=128>"),
    ("conv_down3_kernel<true, false, false, 128, 1>", "conv_down3_kernel<NW=128,NORM>"),
    ("conv_down3_kernel<false, true, false, 64, 2>", "conv_down3_kernel<NW=64,BWDNORM>"),
    ("conv_down3_kernel<false, true, false, 64, 0>", "conv_down3_kernel<NW=64>"),
    ("_ZN12_GLOBAL__N_117conv_down3_kernelILb0ELb1ELb0ELi64ELi2EEEvNS_8D3ParamsE", "conv_down3_kernel<NW=64,BWDNORM>"),
    ("_ZN12_GLOBAL__N_117conv_down3_kernelILb1ELb0ELb0ELi128ELi1EEEvNS_8D3ParamsE", "conv_down3_kernel<NW=128,NORM>"),
    ("_ZN12_GLOBAL__N_119s1t_fwd_rows_kernelILi32ELb1ELi2EEEvPKDF16bPKfS4_PfiiiiNS_10RowsNormInE", "s1t_fwd_rows_kernel<32,NORM>"),
    ("n3_wgrad16_kernel<1, 16>", "n3_wgrad16_kernel<1,16>"),
    ("conv_up3_kernel<128, 64, false, true>", "conv_up3_kernel<128,64>"),
    ("void (anonymous namespace)::conv_up3_kernel<64, 32, true, false>((anonymous namespace)::U3Params)", "conv_up3_kernel<64,32>"),
    ("conv_up3_kernel<64, 32, true, false, 1>", "conv_up3_kernel<64,32,4w>"),
    ("conv_up3_kernel<64, 32, true, false, 2>", "conv_up3_kernel<64,32>"),
    ("conv_up3_kernel<128, 64, false, true, 1>", "conv_up3_kernel<128,64>"),
    ("conv_up4_kernel<true, false>", "conv_up4_kernel"),
    ("conv_up4_kernel<false, true, true>", "conv_up4_kernel<PAIR>"),
    ("conv_up4_kernel<true, false, false>", "conv_up4_kernel"),
    ("wgrad_at_kernel<16, 8>", "wgrad_at_kernel<16,8>"),
    ("wgrad_at32_kernel<16, 4>", "wgrad_at32_kernel<16,4>"),
    ("patch_p16_kernel<1, 32, true, false, true>", "patch_p16_kernel<1,32,nf>"),
    ("patch_p16_kernel<2, 64, true, true, false>", "patch_p16_kernel<2,64>"),
    ("_ZN12_GLOBAL__N_119s1t_fwd_rows_kernelILi32ELb0EEEvPKDF16bPKfS4_PfiiiiNS_10RowsNormInE", "s1t_fwd_rows_kernel<32>"),
    ("n3_wgrad16_kernel<2>", "n3_wgrad16_kernel<2>"),
    ("up_p16_kernel<64>", "up_p16_kernel"),
    ("_ZN12_GLOBAL__N_118bwd_apply16_kernelILb0ELb1EEEvPKDF16bPKvPKfS6_PfPDF16bxxiifS7_i", "bwd_apply16_kernel<false,true>"),
    ("apply16_kernel<0>", "apply16_kernel<0>"),
    ("bwd_final_kernel", "bwd_final_kernel"),
]


def test_short_names():
    for raw, want in CASES:
        assert short(raw) == want, (raw, short(raw), want)
