"""CPU tier: the GNU Radio side of the drop-in boundary, as far as it can be checked in an image without GNU Radio.

  * host/jrc_blocks.{h,cc} — the code that is compiled into gnuradio-mimo_ofdm_jrc with -DJRC_WITH_GNURADIO — must not lean on
    anything only the stand-alone test runtime has (the t_* capture hooks, pmt::to_json): those belong to jrc_blocks_capi.cc;
  * every block's make() has the reference's parameter list (checked against /root/reference/include when that tree is present, i.e.
    in the build container), and the reference's GRC descriptors call make() with that many arguments;
  * gr/CMakeLists.txt builds every kernel source build.py builds; gr/swig wraps every block class the header declares;
  * with GNU Radio 3.8 development files present (pkg-config gnuradio-runtime) the GR branch is compiled with -fsyntax-only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "host")
REF = "/root/reference"


def _read(*p):
    return open(os.path.join(*p)).read()


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _make_params(header_text, cls):
    """parameter TYPES of `static sptr make(...)` of class `cls`, whitespace- and name-insensitive"""
    m = re.search(r"class\s+(?:\w+\s+)?%s\s*:[^{]*\{(.*?)\n\s*\};" % cls, header_text, flags=re.S)
    assert m, cls
    mk = re.search(r"static\s+sptr\s+make\s*\((.*?)\)\s*;", m.group(1), flags=re.S)
    assert mk, cls
    out = []
    depth, cur = 0, ""
    for ch in mk.group(1):
        if ch in "<(":
            depth += 1
        if ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    types = []
    for a in out:
        a = a.split("=")[0].strip()
        a = re.sub(r"\s+", " ", a)
        a = re.sub(r"\s*([&*<>,])\s*", r"\1", a)
        t = re.sub(r"(\w+)$", "", a).strip()             # drop the parameter name
        t = t.replace("std::", "")
        types.append(t)
    return types


def test_gr_branch_uses_no_test_runtime_extras():
    for f in ("jrc_blocks.cc", "jrc_blocks.h"):
        src = _strip_comments(_read(HOST, f))
        assert not re.search(r"\bt_(out_tags|published|read|written|in_tags|msgs)\b", src), f
        assert "to_json" not in src, f
        assert "jrc_host::" not in src, f                # the harness namespace
    capi = _read(HOST, "jrc_blocks_capi.cc")
    assert "#ifndef JRC_WITH_GNURADIO" in capi           # the harness compiles to nothing in a GNU Radio build


BLOCKS = ["mimo_ofdm_radar", "matrix_transpose", "range_angle_estimator", "ofdm_cyclic_prefix_remover", "fft_peak_detect",
          "mimo_ofdm_equalizer", "mimo_precoder", "target_simulator", "stream_encoder", "stream_decoder", "moving_avg", "frame_detector",
          "frame_sync", "zero_pad", "ofdm_frame_generator"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="reference tree not present (GPU box): signatures are compared in the build container")
@pytest.mark.parametrize("blk", BLOCKS)
def test_make_signature_equals_the_reference(blk):
    ours = _make_params(_strip_comments(_read(HOST, "jrc_blocks.h")), blk)
    ref = _make_params(_strip_comments(_read(REF, "include", "mimo_ofdm_jrc", blk + ".h")), blk)
    assert ours == ref, (blk, ours, ref)
    yml = _read(REF, "grc", "mimo_ofdm_jrc_%s.block.yml" % blk)
    mk = re.search(r"make:\s*(?:\|-?\s*)?mimo_ofdm_jrc\.%s\((.*?)\)\s*$" % blk, yml, flags=re.S | re.M)
    if mk:                                               # the flowgraph-side call: as many arguments as make() takes (defaults aside)
        n_call = len(re.findall(r"\$\{", mk.group(1)))
        assert n_call <= len(ours), (blk, n_call, len(ours))


def _class_body(header_text, cls):
    m = re.search(r"class\s+(?:\w+\s+)?%s\s*:[^{]*\{(.*?)\n\s*\};" % cls, header_text, flags=re.S)
    assert m, cls
    return m.group(1)


def _virtuals(header_text, cls):
    """the pure-virtual members of class `cls` (setters / getters of the public interface) as normalised 'ret name(types) [const]' strings"""
    out = []
    for ret, name, args, const in re.findall(r"virtual\s+([\w:<>\s]+?)\s+(\w+)\s*\(([^)]*)\)\s*(const)?\s*=\s*0\s*;", _class_body(header_text, cls)):
        types = []
        for a in [x for x in args.split(",") if x.strip()]:
            a = re.sub(r"\s+", " ", a.split("=")[0].strip())
            a = re.sub(r"\s*([&*<>,])\s*", r"\1", a)
            types.append(re.sub(r"(\w+)$", "", a).strip().replace("std::", ""))
        out.append("%s %s(%s)%s" % (re.sub(r"\s+", " ", ret).replace("std::", ""), name, ",".join(types), " const" if const else ""))
    return sorted(out)


def _enums(header_text):
    """{enum name: [(enumerator, value), ...]} of the global enums a header declares"""
    out = {}
    for name, body in re.findall(r"enum\s+(\w+)\s*(?::\s*\w+\s*)?\{([^}]*)\}", header_text):
        vals, nxt = [], 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            k, _, v = item.partition("=")
            nxt = int(v.strip(), 0) if v.strip() else nxt
            vals.append((k.strip(), nxt))
            nxt += 1
        out[name] = vals
    return out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="reference tree not present (GPU box): compared in the build container")
@pytest.mark.parametrize("blk", BLOCKS)
def test_setters_and_class_decoration_equal_the_reference(blk):
    """beyond make(): every pure-virtual member of the reference's public class (set_*, getters) is declared here with the same
    signature, and the class carries the MIMO_OFDM_JRC_API decoration"""
    ours_h, ref_h = _strip_comments(_read(HOST, "jrc_blocks.h")), _strip_comments(_read(REF, "include", "mimo_ofdm_jrc", blk + ".h"))
    assert _virtuals(ours_h, blk) == _virtuals(ref_h, blk), blk
    assert re.search(r"class\s+MIMO_OFDM_JRC_API\s+%s\s*:" % blk, ours_h) and re.search(r"class\s+MIMO_OFDM_JRC_API\s+%s\s*:" % blk, ref_h)
    base_o = re.search(r"class\s+MIMO_OFDM_JRC_API\s+%s\s*:\s*virtual\s+public\s+(?:jrc_rt|gr)::(\w+)" % blk, ours_h).group(1)
    base_r = re.search(r"class\s+MIMO_OFDM_JRC_API\s+%s\s*:\s*virtual\s+public\s+gr::(\w+)" % blk, ref_h).group(1)
    assert base_o == base_r, (blk, base_o, base_r)         # gr::block / gr::sync_block / gr::tagged_stream_block, as the reference's


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="reference tree not present (GPU box): compared in the build container")
def test_public_enums_equal_the_reference():
    """ChannelEstimator, Modulation (mimo_ofdm_equalizer.h:27-36), MCS, PACKET_TYPE (stream_encoder.h:27-39): same names, same values"""
    ours = _enums(_strip_comments(_read(HOST, "jrc_blocks.h")))
    ref = {}
    for f in ("mimo_ofdm_equalizer.h", "stream_encoder.h"):
        ref.update(_enums(_strip_comments(_read(REF, "include", "mimo_ofdm_jrc", f))))
    assert set(ref) == {"ChannelEstimator", "Modulation", "MCS", "PACKET_TYPE"}
    for name, vals in ref.items():
        assert ours.get(name) == vals, (name, ours.get(name), vals)
    hdr = _read(HOST, "jrc_blocks.h")
    assert "#define MIMO_OFDM_JRC_API __GR_ATTR_EXPORT" in hdr and "gnuradio_mimo_ofdm_jrc_EXPORTS" in hdr     # api.h:27-31


def test_cmake_and_swig_cover_what_the_tree_builds():
    import importlib
    jb = importlib.import_module("gr-mimo-ofdm-jrc_amd.build")
    cm = _read(ROOT, "gr", "CMakeLists.txt")
    listed = re.search(r"set\(JRC_HIP_SOURCES ([^)]*)\)", cm).group(1).split()
    assert sorted(listed) == sorted(os.path.splitext(s)[0] for s in jb.SOURCES)
    for flag in ("-O3", "-std=c++17", "-fno-gpu-rdc"):
        assert flag in cm and flag in jb.HIPCC_FLAGS
    assert "JRC_WITH_GNURADIO" in cm and "gnuradio-mimo_ofdm_jrc" in cm
    hdr = _strip_comments(_read(HOST, "jrc_blocks.h"))
    classes = re.findall(r"class\s+(?:MIMO_OFDM_JRC_API\s+)?(\w+)\s*:\s*virtual\s+public\s+jrc_rt::", hdr)
    swig = _read(ROOT, "gr", "swig", "mimo_ofdm_jrc_swig.i")
    wrapped = re.findall(r"GR_SWIG_BLOCK_MAGIC2\(mimo_ofdm_jrc,\s*(\w+)\)", swig)
    assert sorted(classes) == sorted(wrapped)
    assert set(BLOCKS) <= set(classes)
    import yaml
    y = yaml.safe_load(_read(ROOT, "gr", "grc", "mimo_ofdm_jrc_radar_chain.block.yml"))
    n_params = len(y["parameters"])
    assert y["templates"]["make"].count("${") == n_params == len(_make_params(hdr, "radar_chain"))


def test_gr_branch_compiles_when_gnuradio_is_installed():
    pc = shutil.which("pkg-config")
    if not pc or subprocess.run([pc, "--exists", "gnuradio-runtime"]).returncode != 0:
        pytest.skip("no GNU Radio development files in this image (pkg-config gnuradio-runtime): the -DJRC_WITH_GNURADIO branch of "
                    "host/jrc_blocks.cc cannot be compiled here; gr/CMakeLists.txt builds it where GNU Radio 3.8 is installed")
    flags = subprocess.check_output([pc, "--cflags", "gnuradio-runtime"], text=True).split()
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-DJRC_WITH_GNURADIO", "-I" + HOST, os.path.join(HOST, "jrc_blocks.cc")] + flags,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
