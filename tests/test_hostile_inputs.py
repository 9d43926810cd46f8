"""Hostile input to the host-side native code: truncated, oversized, odd-length and lying RIFF files for afx_wav_probe /
afx_wav_read_s16 (untrusted files, native threads), and hostile offsets / lengths for the clip-record builder.  CPU only.
Run under AddressSanitizer + UBSan with `make -C audio_feature_extraction_amd/csrc asan-test`; in the
ordinary suite the same cases check the status codes."""
import os
import struct

import numpy as np
import pytest

from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd import wavio


def _fmt(tag=1, ch=1, rate=22050, bits=16, size=16, extra=b""):
    body = struct.pack("<HHIIHH", tag, ch, rate, (rate * ch * bits // 8) & 0xFFFFFFFF, (ch * bits // 8) & 0xFFFF, bits) + extra
    return b"fmt " + struct.pack("<I", size) + body


def _riff(chunks: bytes, riff_size=None):
    return b"RIFF" + struct.pack("<I", len(chunks) + 4 if riff_size is None else riff_size) + b"WAVE" + chunks


def _data(payload: bytes, size=None):
    return b"data" + struct.pack("<I", len(payload) if size is None else size) + payload


PCM = (np.arange(1000, dtype=np.int16) * 7 - 3000).tobytes()

CASES = {
    # name: (bytes, probe status, frames when ok)
    "empty": (b"", 1, 0),
    "four_bytes": (b"RIFF", 1, 0),
    "riff_only": (b"RIFF\x04\x00\x00\x00WAVE", 1, 0),
    "not_wave": (b"RIFF\x04\x00\x00\x00AVI " + _fmt() + _data(PCM), 1, 0),
    "truncated_in_fmt": (_riff(_fmt())[:20], 1, 0),
    "fmt_too_small": (_riff(b"fmt " + struct.pack("<I", 8) + b"\x01\x00\x01\x00\x22\x56\x00\x00" + _data(PCM)), 1, 0),
    "oversized_fmt": (_riff(_fmt(size=0x7FFFFFF0) + _data(PCM)), 1, 0),          # the fmt chunk claims 2 GB: no data chunk follows
    "fmt_4gb": (_riff(_fmt(size=0xFFFFFFFF) + _data(PCM)), 1, 0),
    "data_before_fmt": (_riff(_data(PCM) + _fmt()), 1, 0),
    "data_claims_4gb": (_riff(_fmt() + _data(PCM, size=0xFFFFFFFF)), 0, 1000),    # cut at the file's end
    "data_odd_length": (_riff(_fmt() + _data(PCM + b"\x7f")), 0, 1000),          # a trailing half sample is dropped
    "data_one_byte": (_riff(_fmt() + _data(b"\x01")), 0, 0),
    "data_truncated_header": (_riff(_fmt() + b"data\x10\x00"), 1, 0),
    "zero_channels": (_riff(_fmt(ch=0) + _data(PCM)), 1, 0),
    "zero_bits": (_riff(_fmt(bits=0) + _data(PCM)), 1, 0),
    "huge_channels": (_riff(_fmt(ch=65535, bits=65528) + _data(PCM)), 0, 0),
    "odd_chunk_padding": (_riff(b"LIST" + struct.pack("<I", 3) + b"abc\x00" + _fmt() + _data(PCM)), 0, 1000),
    "junk_chunk_4gb": (_riff(b"JUNK" + struct.pack("<I", 0xFFFFFFFF) + b"xx" + _fmt() + _data(PCM)), 1, 0),
    "extensible": (_riff(_fmt(tag=0xFFFE, size=40, extra=struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + b"\x00" * 14)
                         + _data(PCM)), 0, 1000),
    "extensible_short": (_riff(_fmt(tag=0xFFFE, size=18, extra=b"\x00\x00") + _data(PCM)), 0, 1000),
    "good": (_riff(_fmt() + _data(PCM)), 0, 1000),
}


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostile")
    out = {}
    for name, (blob, _, _) in CASES.items():
        p = d / (name + ".wav")
        p.write_bytes(blob)
        out[name] = str(p)
    out["missing"] = str(d / "does_not_exist.wav")
    return out


def test_probe_classifies_every_hostile_header(files):
    names = list(CASES) + ["missing"]
    for threads in (1, 4):
        pr = N.wav_probe([files[k] for k in names], threads)
        for i, k in enumerate(names):
            want_status, want_frames = (2, 0) if k == "missing" else CASES[k][1:]
            assert pr["status"][i] == want_status, (k, int(pr["status"][i]))
            if want_status == 0:
                assert pr["frames"][i] == want_frames, (k, int(pr["frames"][i]))
                assert 0 < pr["data_off"][i] <= os.path.getsize(files[k])
            else:
                assert pr["frames"][i] == 0 and pr["data_off"][i] == 0


def test_probe_and_python_decoder_agree_on_what_is_a_wave_file(files):
    """The native walker follows wavio._parse: a file one rejects the other rejects."""
    for k, (_, status, frames) in CASES.items():
        pr = N.wav_probe([files[k]], 1)
        try:
            a, rate, kind = wavio.read_wav_raw(files[k])
            ok = True
        except Exception:
            ok = False
        if pr["status"][0] == 0 and pr["tag"][0] == 1 and pr["bits"][0] == 16 and pr["channels"][0] == 1:
            assert ok and a.shape[0] == pr["frames"][0], k
        if pr["status"][0] == 1 and k not in ("zero_channels", "zero_bits"):
            assert not ok, k


def test_read_s16_never_writes_outside_its_slot(files):
    names = ["good", "data_claims_4gb", "data_odd_length", "odd_chunk_padding", "extensible"]
    paths = [files[k] for k in names]
    pr = N.wav_probe(paths, 2)
    lens = pr["frames"].astype(np.int64)
    offs = np.arange(len(names), dtype=np.int64) * 1100 + 50
    out = np.full(len(names) * 1100 + 100, 12345, np.int16)
    st = N.wav_read_s16(paths, pr["data_off"], lens, out, offs, 3)
    assert (st == 0).all()
    ref = np.frombuffer(PCM, np.int16)
    mask = np.ones(out.size, bool)
    for o, ln in zip(offs, lens):
        np.testing.assert_array_equal(out[o:o + ln], ref[:ln])
        mask[o:o + ln] = False
    assert (out[mask] == 12345).all()                                # guard values untouched
    # a file shorter than the caller claims: the read fails, no out-of-bounds write, the other files are read
    st = N.wav_read_s16(paths[:2], pr["data_off"][:2], np.array([1000, 4000], np.int64), out, np.array([0, 1100], np.int64), 2)
    assert st[0] == 0 and st[1] == 2
    # hostile arguments are refused before any thread starts
    for bad_off, bad_len in ((np.array([out.size - 10], np.int64), np.array([100], np.int64)),
                             (np.array([-1], np.int64), np.array([10], np.int64)),
                             (np.array([0], np.int64), np.array([-5], np.int64)),
                             (np.array([2 ** 62], np.int64), np.array([2 ** 62], np.int64)),
                             (np.array([5], np.int64), np.array([2 ** 63 - 1], np.int64))):
        with pytest.raises(ValueError):
            N.wav_read_s16(paths[:1], pr["data_off"][:1], bad_len, out, bad_off, 1)
    st = N.wav_read_s16(paths[:1], np.array([-4], np.int64), np.array([10], np.int64), out, np.array([0], np.int64), 1)
    assert st[0] == 2                                                # a negative file offset is a failed read, not a crash
    st = N.wav_read_s16(paths[:1], np.array([2 ** 62], np.int64), np.array([10], np.int64), out, np.array([0], np.int64), 1)
    assert st[0] == 2


def test_clip_record_builder_refuses_hostile_geometry():
    p = N.make_params(22050, 1024, 256, 13)
    g = N.batch_geometry(p, [0, 1000, 250000], [999, 220500, 0])
    assert g["tmax"].tolist() == [4, 862, 1] and g["tpad"].tolist() == [16, 864, 16]
    assert g["frame_base"].tolist() == [0, 16, 880] and g["blk_base"].tolist() == [0, 1, 55]
    assert (g["frame_slots"], g["blocks"], g["max_tmax"]) == (896, 56, 862)
    assert g["trim_blocks"] == 2 + 431 + 0
    assert N.batch_geometry(p, [], [])["blocks"] == 0
    for offs, lens in (([-1], [10]), ([0], [-10]), ([0], [2 ** 62]), ([2 ** 62], [10]), ([0], [2 ** 63 - 1]),
                       ([0] * 3, [2 ** 40] * 3)):
        with pytest.raises(ValueError):
            N.batch_geometry(p, offs, lens)
    # many maximal clips: the block count overflows the kernels' 32-bit indices -> refused, not wrapped
    n = 40000
    with pytest.raises(ValueError):
        N.batch_geometry(p, np.zeros(n, np.int64), np.full(n, 2 ** 36, np.int64))


def test_host_sources_are_clean_under_asan_and_ubsan():
    """`make asan-test`: the host-only sources rebuilt with g++ -fsanitize=address,undefined, and this file, test_wavio.py
    and test_native_cpu.py run against that library (AFX_LIB).  Skipped inside that run itself and where g++ has no
    sanitizer runtime."""
    import shutil
    import subprocess
    if os.environ.get("AFX_LIB", "").endswith("libafx_host_asan.so"):
        pytest.skip("already running under the sanitizer build")
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("g++ has no AddressSanitizer runtime here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("AFX_LIB", "LD_PRELOAD")}
    r = subprocess.run(["make", "-C", os.path.join(root, "audio_feature_extraction_amd", "csrc"), "asan-test"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stdout + r.stderr and "runtime error" not in r.stdout + r.stderr
