"""The roofline block of bench.py names a HIP kernel SYMBOL, spelled exactly as the rocprofv3 summaries and the counter
tables under profiles/ spell it (mggan/hip/ksym.py).  CPU side: the spelling itself, that every launch of the library goes
through the logged launch macro, and that no committed table of this round merges two instantiations."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))

from mggan.hip import ksym  # noqa: E402


def test_short_keeps_every_template_argument():
    cases = {
        "_Z11attn_kernelILi16ELb1EEviPKfS1_S1_S1_S1_S1_S1_PfiS1_iS1_S2_S2_Pd8BnBwdFinPKi.kd": "attn_kernel<16,true>",
        "_Z11attn_kernelILi8ELb0EEviPKfS1_S1_S1_S1_S1_S1_PfiS1_iS1_S2_S2_Pd8BnBwdFinPKi.kd": "attn_kernel<8,false>",
        "_Z22social_rows_bwd_kernelILi32ELi2ELi8ELb1ELi1EEv11SocRowsArgs.kd": "social_rows_bwd_kernel<32,2,8,true,1>",
        "_Z15lstm_bwd_kernelILi64ELi64EEv10SeqBwdArgs": "lstm_bwd_kernel<64,64>",
        "_Z23decoder_bwd_pair_kernel12DecFusedArgs.kd": "decoder_bwd_pair_kernel",
        "_Z21comm_allreduce_kernelIfEv8CommArgs": "comm_allreduce_kernel<float>",
        "_Z19wgrad_stream_kernelILi2EEv11StreamBatch.kd": "wgrad_stream_kernel<2>",
        "_Z11gemm_kernelILb0ELb1ELb1EEv8GemmArgs": "gemm_kernel<false,true,true>",
        "sample_slots_scan_kernel.kd": "sample_slots_scan_kernel",
        "__amd_rocclr_copyBuffer.kd": "__amd_rocclr_copyBuffer",
    }
    for sym, want in cases.items():
        assert ksym.short(sym) == want, (sym, ksym.short(sym))
    assert ksym.family("attn_kernel<16,true>") == "attn_kernel"


def test_launch_log_parsing_and_primary_kernel():
    text = "_Z17image_gram_kerneliPKfPdPKi:65536;_Z26image_gram_finalize_kernelPKdiPd:24576"
    launches = ksym.parse_launch_log(text)
    assert launches == [("image_gram_kernel", 65536), ("image_gram_finalize_kernel", 24576)]
    assert ksym.primary(launches) == "image_gram_kernel"
    assert ksym.primary([]) is None and ksym.parse_launch_log("") == []
    # the first of equals
    assert ksym.primary([("a", 4), ("b", 4)]) == "a"


def test_every_launch_goes_through_the_logged_macro():
    """A kernel launched with a bare hipLaunchKernelGGL / <<<>>> would be invisible to the launch log: bench.py could then
    book an entry's time on the wrong symbol."""
    csrc = os.path.join(ROOT, "mg-gan_amd", "csrc")
    n = 0
    for path in glob.glob(os.path.join(csrc, "*.hip")):
        src = open(path).read()
        assert "hipLaunchKernelGGL" not in src and "<<<" not in src, path
        n += len(re.findall(r"\bMG_LAUNCH\(", src))
    assert n > 100
    common = open(os.path.join(csrc, "common.h")).read()
    assert common.count("hipLaunchKernelGGL(") == 1 and "mggan_note_launch" in common


def test_bench_names_kernels_by_logged_symbol():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_symbols", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    log = ksym.parse_launch_log("_Z11attn_kernelILi8ELb1EEviPKf:196608;_Z18partial_sum_kernelPKfiiPf:4096")
    assert B.kernel_of("mggan_scene_attention_bwd", log) == "attn_kernel<8,true>"
    assert not B.kernel_of("mggan_scene_attention_bwd", log).startswith("mggan_")
    assert not B.kernel_of("mggan_wgrad_multi", []).startswith("mggan_")


def test_committed_counter_tables_of_this_round_keep_instantiations_apart():
    """profiles/hbm_traffic_c*.json and mfma_util_c*.json (regenerated every round by tools/profile_round.sh): a key such as
    attn_kernel<16> would be the forward and the adjoint kernel summed."""
    for name in ("hbm_traffic_c2.json", "hbm_traffic_c3.json", "mfma_util_c2.json", "mfma_util_c3.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        tab = json.load(open(path))
        attn = sorted(k for k in tab if k.startswith("attn_kernel"))
        assert attn and all(re.fullmatch(r"attn_kernel<\d+,(true|false)>", k) for k in attn), (name, attn)
