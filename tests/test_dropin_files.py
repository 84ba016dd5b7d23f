"""The installable drop-in artefacts (VERDICT r1 'missing' #1): GRC descriptor, SWIG stanza,
HIP-gated CMake project.  CPU only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def test_grc_descriptor_declares_the_reference_surface():
    """Self-contained check (runs anywhere): key, import, make/callback templates, parameter keys and defaults,
    ports -- grc/baz_music_doa.xml:4-8,10-97 of the reference."""
    import xml.etree.ElementTree as ET
    blk = ET.parse(os.path.join(ROOT, "grc", "baz_music_doa.xml")).getroot()
    assert blk.findtext("key") == "baz_music_doa"
    assert blk.findtext("import") == "from baz import music_doa_helper"
    assert blk.findtext("make") == ("music_doa_helper.music_doa_helper(m=$m, n=$n, nsamples=$nsamples, "
                                    "angular_resolution=$angular_resolution, frequency=$freq, array_spacing=$spacing, "
                                    "antenna_array=$antenna_array, output_spectrum=$output_spectrum)")
    assert blk.findtext("callback") == "set_frequency($freq)"
    params = {p.findtext("key"): p.findtext("value") for p in blk.findall("param")}
    assert params == {"m": "4", "n": "1", "nsamples": "512", "angular_resolution": "360", "freq": "1", "spacing": "1",
                      "antenna_array": "[[0,0],[1,0],[2,0],[3,0]]", "output_spectrum": "False"}
    assert [(s.findtext("name"), s.findtext("type"), s.findtext("vlen")) for s in blk.findall("sink")] == \
        [("in", "complex", "$nsamples")]
    assert [(s.findtext("name"), s.findtext("type"), s.findtext("vlen"), s.findtext("optional")) for s in blk.findall("source")] == \
        [("ang", "float", "$n", None), ("lvl", "float", "$n", None), ("spectrum", "float", "$angular_resolution", "1")]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_grc_descriptor_is_byte_identical_to_the_reference():
    assert _read(os.path.join(ROOT, "grc", "baz_music_doa.xml")) == _read(os.path.join(REF, "grc", "baz_music_doa.xml"))


def test_block_tree_entry_places_the_block_where_gr_baz_does():
    """grc/baz_block_tree.xml of the reference lists baz_music_doa under [gr-baz] / DOA; the stand-alone entry (opt-in at
    install time, a host with gr-baz already has it) names the same path."""
    import xml.etree.ElementTree as ET
    top = ET.parse(os.path.join(ROOT, "grc", "baz_music_block_tree.xml")).getroot()
    assert top.tag == "cat" and top.findtext("name") == "[gr-baz]"
    cats = top.findall("cat")
    assert [(c.findtext("name"), [b.text for b in c.findall("block")]) for c in cats] == [("DOA", ["baz_music_doa"])]
    cm = open(os.path.join(ROOT, "CMakeLists.txt")).read()
    assert re.search(r'option\(BAZ_MUSIC_INSTALL_BLOCK_TREE "[^"]*" OFF\)', cm) and "grc/baz_music_block_tree.xml" in cm
    if os.path.isdir(REF):
        ref = ET.parse(os.path.join(REF, "grc", "baz_block_tree.xml")).getroot()
        assert ref.findtext("name") == "[gr-baz]"
        doa = [c for c in ref.iter("cat") if c.findtext("name") == "DOA"]
        assert len(doa) == 1 and [b.text for b in doa[0].findall("block")] == ["baz_music_doa"]


def _stanza(text, guard):
    """The declarations between `#ifdef <guard>` and its `#endif`, whitespace-normalised, comments dropped."""
    found = [b for b in re.findall(r"#ifdef\s+%s\b(.*?)#endif\s*//\s*%s" % (guard, guard), text, re.S)
             if "GR_SWIG_BLOCK_MAGIC" in b]           # (the reference also guards the header #include with it)
    assert len(found) == 1, "no single #ifdef %s stanza" % guard
    body = re.sub(r"//[^\n]*", "", found[0])
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [" ".join(l.split()) for l in body.splitlines() if l.strip()]


def test_swig_stanza_exposes_factory_class_and_setter():
    lines = _stanza(open(os.path.join(ROOT, "swig", "baz_music.i")).read(), "BAZ_MUSIC_HIP_FOUND")
    joined = "\n".join(lines)
    assert "GR_SWIG_BLOCK_MAGIC(baz,music_doa)" in joined
    assert "baz_music_doa_sptr baz_make_music_doa(unsigned int m, unsigned int n, unsigned int nsamples, const std::vector<std::vector<gr_complex> >& array_response, unsigned int resolution);" in joined
    assert "class baz_music_doa : public gr::sync_block" in joined
    assert "void set_array_response(const std::vector<std::vector<gr_complex> >& array_response);" in joined


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_swig_stanza_is_the_reference_stanza_with_the_guard_swapped():
    ours = _stanza(open(os.path.join(ROOT, "swig", "baz_music.i")).read(), "BAZ_MUSIC_HIP_FOUND")
    ref = _stanza(open(os.path.join(REF, "swig", "baz_swig.i")).read(), "ARMADILLO_FOUND")
    extra = [l for l in ours if l not in ref]
    # ours = the reference's declarations + the include of the block header + the opt-in set_peak_mode extension
    assert [l for l in ref if l not in ours] == []
    assert all(("baz_music_doa.h" in l) or l in ("%{", "%}") or "set_peak_mode" in l for l in extra), extra


def _cmake():
    exe = shutil.which("cmake")
    if not exe:
        pytest.skip("cmake not installed")
    return exe


def test_cmake_project_configures_up_to_gnuradio(tmp_path):
    """Default configuration: the HIP gate passes (kernels will be compiled for gfx950) and the ONLY failure is
    find_package(Gnuradio) -- GNU Radio is not installed here."""
    r = subprocess.run([_cmake(), "-S", ROOT, "-B", str(tmp_path / "b")], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert "compiling MUSIC DOA estimator block for gfx950" in out, out[-2000:]
    assert r.returncode != 0
    errors = re.findall(r"CMake Error at ([^\n]+)", out)
    assert len(errors) == 1 and "find_package" in errors[0], errors
    assert re.search(r'package configuration file provided by\s+"Gnuradio"', out), out[-2000:]


def test_cmake_project_configures_fully_on_the_api_stand_in(tmp_path):
    r = subprocess.run([_cmake(), "-S", ROOT, "-B", str(tmp_path / "b"), "-DBAZ_MUSIC_WITH_GR_SHIM=ON"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    cfg = open(tmp_path / "b" / "config.h").read()
    assert "#define BAZ_MUSIC_HIP_FOUND 1" in cfg
    if os.environ.get("BAZ_TEST_CMAKE_BUILD") == "1":          # ~2 min of hipcc: opt-in
        b = subprocess.run([_cmake(), "--build", str(tmp_path / "b"), "-j", "8"], capture_output=True, text=True, timeout=1800)
        assert b.returncode == 0, (b.stdout + b.stderr)[-3000:]
        for lib in ("libbaz_music_hip.so", "libbaz_agc_hip.so", "libbaz_resamp_hip.so", "libgnuradio-baz-music.so"):
            assert os.path.exists(tmp_path / "b" / lib)


def test_cmake_gate_rejects_other_architectures(tmp_path):
    r = subprocess.run([_cmake(), "-S", ROOT, "-B", str(tmp_path / "b"), "-DCMAKE_HIP_ARCHITECTURES=gfx90a",
                        "-DBAZ_MUSIC_WITH_GR_SHIM=ON"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "gfx950 (MI355X) only" in (r.stdout + r.stderr)
