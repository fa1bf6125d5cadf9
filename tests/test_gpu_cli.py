"""End-to-end drop-in check on the GPU box: the C++ `tiebrush` / `tiecov` command lines (host BAM codec +
libtbk.so) reproduce the reference's golden outputs through the SURVEY.md §4.4 normaliser."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, sample_paths, read_lines

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tiebrush_amd", "_build")


def _run(args, **kw):
    return subprocess.run(args, check=True, capture_output=True, text=True, **kw)


def _compare_bam(out_path, gold_path):
    from tiebrush_amd import bamio
    o, g = bamio.read_bam(out_path, keep_aux=True), bamio.read_bam(gold_path)
    assert o.n == g.n
    for i in range(g.n):
        assert bamio.record_identity(o, i) == bamio.record_identity(g, i), i
        gyc = g.yc[i] if g.has_yc[i] else 1.0
        assert o.has_yc[i] and o.yc[i] == gyc and o.yx[i] == g.yx[i] and o.yd[i] == g.yd[i], i
    return o


@pytest.mark.parametrize("name", ["t1", "t2"])
def test_tiebrush_cli_on_samples(tmp_path, name):
    from tiebrush_amd import bamio
    out = str(tmp_path / "o.bam")
    r = _run([os.path.join(BIN, "tiebrush"), "-A", "-o", out] + sample_paths(name))
    gold = os.path.join(GOLDEN, name, name + ".bam")
    o = _compare_bam(out, gold)
    n_in = {"t1": 416922, "t2": 242910}[name]
    assert "%d input records written as %d" % (n_in, o.n) in r.stderr
    # HEAD tag format: YC:f always, YX always, YD only when > 0, appended in that order on fresh records
    aux = bamio.record_aux(o, 0)
    tags = [t for t, _, _ in aux]
    assert tags[tags.index("YC"):][:2] == ["YC", "YX"] and dict((t, ty) for t, ty, _ in aux)["YC"] == "f"
    assert all(("YD" in t) == (o.yd[i] > 0) for i, t in enumerate(o.aux_types) for t in [set(x.decode() for x in t)])
    hdr = o.header
    assert hdr.is_tiebrush() and len(hdr.co_samples()) == 10 and hdr.co_samples()[0].endswith(name + "s0.bam")


def _normalise_sam_line(line):
    """SURVEY.md §4.4 on one `samtools view` line: -> (the eleven fixed fields + every other tag as text, in order; YC; YX; YD)
    with the absent tags at their defaults (YC 1, YX 1, YD 0) and YC compared by value (golden: YC:i, HEAD: YC:f)"""
    f = line.rstrip("\n").split("\t")
    vals = {"YC": 1.0, "YX": 1, "YD": 0}
    rest = []
    for a in f[11:]:
        tag, ty, v = a.split(":", 2)
        if tag in vals:
            assert ty in ("i", "f"), a
            vals[tag] = float(v) if tag == "YC" else int(v)
        else:
            rest.append(a)
    return tuple(f[:11] + rest), vals["YC"], vals["YX"], vals["YD"]


def test_tiebrush_cli_t2_equals_the_reference_sam_text(tmp_path):
    """The reference keeps the text form of golden t2 (`test/t2/t2.sam`, what its run_tests.sh diffs with `samtools view`): the
    command line's output, rendered as SAM text, equals it line for line through the §4.4 normaliser — every fixed field, every
    carried tag with its type and text, their order, and the three counters by value."""
    import samtext
    out = str(tmp_path / "o.bam")
    _run([os.path.join(BIN, "tiebrush"), "-A", "-o", out] + sample_paths("t2"))
    txt, _ = samtext.bam_to_sam_text(out)
    mine = [ln for ln in txt.splitlines(True) if not ln.startswith("@")]
    gold = open(os.path.join(GOLDEN, "t2", "t2.sam")).read().splitlines(True)
    assert len(mine) == len(gold) == 8179
    for i, (a, b) in enumerate(zip(mine, gold)):
        assert _normalise_sam_line(a) == _normalise_sam_line(b), i
    # HEAD's tag text: YC:f always, YX:i always, YD:i only when positive, appended in that order behind the record's own tags
    f = mine[1].rstrip("\n").split("\t")
    assert [a[:5] for a in f if a[:2] in ("YC", "YX", "YD")][:2] == ["YC:f:", "YX:i:"]


@pytest.mark.parametrize("name", ["t1", "t2"])
@pytest.mark.parametrize("flag", ["-P", "--exon"])
def test_tiebrush_cli_clip_exon_equal_golden(tmp_path, name, flag):
    """-P / -E on M/N-only fixtures reproduce the default-mode golden BAMs"""
    out = str(tmp_path / "o.bam")
    _run([os.path.join(BIN, "tiebrush"), "-A", flag, "-o", out] + sample_paths(name))
    _compare_bam(out, os.path.join(GOLDEN, name, name + ".bam"))


@pytest.mark.parametrize("flag", ["--clip", "-E"])
def test_tiebrush_cli_recollapse_clip_exon(tmp_path, flag):
    from tiebrush_amd import bamio
    out = str(tmp_path / "t12.bam")
    _run([os.path.join(BIN, "tiebrush"), "-A", flag, "-o", out, os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")])
    o, g = bamio.read_bam(out), bamio.read_bam(os.path.join(GOLDEN, "t12.bam"))
    assert o.n == g.n == 9491
    for i in range(g.n):
        assert bamio.record_identity(o, i) == bamio.record_identity(g, i)
        assert o.yx[i] == g.yx[i] and o.yd[i] == g.yd[i]


def test_tiebrush_cli_recollapse_and_listfile(tmp_path):
    lst = tmp_path / "inputs.txt"
    lst.write_text("# list of inputs\n%s\n%s\n" % (os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")))
    out = str(tmp_path / "t12.bam")
    _run([os.path.join(BIN, "tiebrush"), "--collapse-same", "-o", out, str(lst)])
    from tiebrush_amd import bamio
    o, g = bamio.read_bam(out), bamio.read_bam(os.path.join(GOLDEN, "t12.bam"))
    assert o.n == g.n == 9491
    for i in range(g.n):
        assert bamio.record_identity(o, i) == bamio.record_identity(g, i)
        assert o.yx[i] == g.yx[i] and o.yd[i] == g.yd[i]
        # inputs carry integer YC (0.0.6): the float update fails in htslib and the stale value survives on records that
        # already had the tag; records without it get the float.  Value parity therefore holds for the latter only.
    assert len(o.header.co_samples()) == 20


@pytest.mark.parametrize("name", ["t1", "t2"])
def test_tiecov_cli(tmp_path, name):
    pre = str(tmp_path / name)
    _run([os.path.join(BIN, "tiecov"), "-s", pre + ".sample", "-c", pre + ".coverage", "-j", pre + ".junctions",
          os.path.join(GOLDEN, name, name + ".bam")])

    def norm(lines, col):
        out = []
        for l in lines:
            f = l.split("\t")
            if len(f) > col:
                assert f[col].endswith(".000"), l
                f[col] = f[col][:-4]
            out.append("\t".join(f))
        return out

    assert norm(read_lines(pre + ".coverage.bedgraph"), 3) == read_lines(os.path.join(GOLDEN, name, name + ".coverage.bedgraph"))
    assert norm(read_lines(pre + ".junctions.bed"), 4) == read_lines(os.path.join(GOLDEN, name, name + ".junctions.bed"))
    ours = read_lines(pre + ".sample.bedgraph")
    gold = read_lines(os.path.join(GOLDEN, name, name + ".sample.bedgraph"))
    assert ours[0] == gold[0]
    assert ["\t".join(l.split("\t")[:4]) for l in ours[1:]] == ["\t".join(l.split("\t")[:4]) for l in gold[1:]]
    for l in ours[1:]:
        f = l.split("\t")
        want = np.float32(np.float32(np.float32(int(f[3])) / np.float32(10)) * (np.float32(1.5) - np.float32(0.1))) + np.float32(0.1)
        assert f[4] == "%f" % want


def test_cli_usage_errors():
    r = subprocess.run([os.path.join(BIN, "tiecov"), "-h"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout
    r = subprocess.run([os.path.join(BIN, "tiebrush"), os.path.join(GOLDEN, "t12.bam")], capture_output=True, text=True)
    assert r.returncode == 1 and "output filename must be provided" in r.stderr
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "-L", "-P", "-o", "/tmp/x.bam", os.path.join(GOLDEN, "t12.bam")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "only one merging strategy" in r.stderr


@pytest.mark.parametrize("profile,flags,okw", [
    ("c5", ["-E", "-N", "5", "-Q", "1"], dict(strategy=3, max_nh=5, min_qual=1)),
    ("c3", ["--clip"], dict(strategy=2)),
    ("c5", ["--keep-secondary", "-S", "--store-frac"], dict(keep_secondary=True, keep_supplementary=True, store_frac=True)),
])
def test_cli_on_synthetic_bams_matches_oracle(tmp_path, profile, flags, okw):
    """options no reference fixture exercises (-E/-P/-N/-Q/-S/--store-frac), end to end through real BAM files:
    tiebrush CLI output == oracle on the same records; tiecov CLI text == oracle coverage of that output"""
    import struct
    from oracle import oracle_ffi as orc
    from tiebrush_amd import bamio, synth, soa
    tile = synth.make_tile(3, 4000, profile, n_loci=60)
    paths = synth.write_bams(tile, str(tmp_path / "in"))
    out = str(tmp_path / "out.bam")
    r = _run([os.path.join(BIN, "tiebrush"), "-o", out] + flags + paths)
    want = orc.collapse(tile, **okw)
    o = bamio.read_bam(out)
    assert o.n == want["n_groups"]
    assert "%d input records written as %d" % (want["n_passed"], want["n_groups"]) in r.stderr
    fo = tile.file_of()
    for g in range(o.n):
        gi = int(want["rep"][g])
        f = int(fo[gi])
        assert o.qname[g] == b"r%d_%d" % (f, gi - int(tile.file_off[f]))
        assert np.float32(o.yc[g]) == np.float32(want["yc"][g]) and o.yx[g] == want["yx"][g] and o.yd[g] == want["yd"][g]
    pre = str(tmp_path / "cov")
    _run([os.path.join(BIN, "tiecov"), "-c", pre, "-j", pre, out])
    cw = orc.coverage(soa.cov_input_from_bam(o))
    names = o.header.ref_names
    want_cov = ["track type=bedGraph"] + ["%s\t%d\t%d\t%.3f" % (names[cw["iv_tid"][i]], cw["iv_start"][i], cw["iv_end"][i], cw["iv_val"][i])
                                          for i in range(cw["n_intervals"])]
    want_j = ["track name=junctions"] + ["%s\t%d\t%d\tJUNC%08d\t%.3f\t%s" % (names[cw["j_tid"][i]], cw["j_start"][i], cw["j_end"][i], i + 1,
                                                                            cw["j_val"][i], chr(cw["j_strand"][i]))
                                         for i in range(cw["n_junctions"])]
    assert read_lines(pre + ".bedgraph") == want_cov
    assert read_lines(pre + ".bed") == want_j


def test_cli_tiling_by_reference_is_exact(tmp_path):
    """TBK_TILE_RECORDS forces one tile per reference sequence: byte-identical output BAM to the single-tile run"""
    from tiebrush_amd import bamio
    a, b = str(tmp_path / "one.bam"), str(tmp_path / "tiled.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", a] + sample_paths("t2"))
    env = dict(os.environ, TBK_TILE_RECORDS="1000", TBK_DEVICE_DECODE="0")
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "-o", b] + sample_paths("t2"), check=True, capture_output=True, text=True, env=env)
    assert "242910 input records written as 8179" in r.stderr
    ra, rb = bamio.bgzf_decompress(open(a, "rb").read()), bamio.bgzf_decompress(open(b, "rb").read())
    ha, pa = bamio.parse_header(ra)
    hb, pb = bamio.parse_header(rb)
    assert ra[pa:] == rb[pb:]
    # several references with reads: synthetic 3-contig input, tiny tiles
    from tiebrush_amd import synth
    tile = synth.make_tile(3, 3000, "c2", n_loci=40)
    paths = synth.write_bams(tile, str(tmp_path / "syn"))
    c, d = str(tmp_path / "c.bam"), str(tmp_path / "d.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", c] + paths)
    subprocess.run([os.path.join(BIN, "tiebrush"), "-o", d] + paths, check=True, capture_output=True, env=dict(os.environ, TBK_TILE_RECORDS="1", TBK_DEVICE_DECODE="0"))
    rc_, rd_ = bamio.bgzf_decompress(open(c, "rb").read()), bamio.bgzf_decompress(open(d, "rb").read())
    assert rc_[bamio.parse_header(rc_)[1]:] == rd_[bamio.parse_header(rd_)[1]:]


def test_cli_decode_paths_give_the_same_record_stream(tmp_path):
    """the five ways the command line can bring its inputs in — whole-input host loader (default), streaming host reader, tiny
    streamed tiles, device decode, hybrid (the GPU and the cores a share of the files each) — the two deflate codecs, and the fall-back from a whole-input tile the GPU refuses (out of
    memory) to the streaming path write the same records byte for byte: plain inputs (tags appended to fresh records) and
    TieBrush-merged inputs (tags updated in place, stale integer YC and all)"""
    from tiebrush_amd import bamio
    cases = {"plain": sample_paths("t2"), "merged": [os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")] + sample_paths("t1")[:2]}
    for name, paths in cases.items():
        streams = {}
        for tag, env in (("whole", {}), ("stream", dict(TBK_HOST_FAST="0")), ("tiles", dict(TBK_TILE_RECORDS="5000")),
                         ("device", dict(TBK_DEVICE_DECODE="1")), ("zlib", dict(TBK_NO_LIBDEFLATE="1")),
                         ("whole_nomem", dict(TBK_TEST_WHOLE_ENOMEM="1")), ("device_nomem", dict(TBK_DEVICE_DECODE="1", TBK_TEST_WHOLE_ENOMEM="1")),
                         ("hybrid", dict(TBK_HYBRID="1")), ("hybrid_80", dict(TBK_HYBRID="1", TBK_HYBRID_SHARE="80")),
                         ("hybrid_nomem", dict(TBK_HYBRID="1", TBK_TEST_WHOLE_ENOMEM="1"))):
            out = str(tmp_path / ("%s_%s.bam" % (name, tag)))
            r = subprocess.run([os.path.join(BIN, "tiebrush"), "-o", out] + paths, check=True, capture_output=True, text=True,
                               env=dict(os.environ, TBK_TIMING="1", **env))
            if tag == "whole":
                assert "host path ms" in r.stderr             # the whole-input loader really ran
            if tag == "device":
                assert "device decode:" in r.stderr
            if tag == "whole_nomem":                          # TBK_ENOMEM / TBK_E2BIG on the one big tile: the streaming path takes over
                assert "whole-input tile not used" in r.stderr
            if tag == "device_nomem":
                assert "device decode given up" in r.stderr
            if tag.startswith("hybrid") and name == "plain":   # (the GPU decodes the first files while the cores decode the rest: tbk_tile_join)
                assert ("hybrid decode given up" if tag == "hybrid_nomem" else "hybrid path ms: device") in r.stderr, r.stderr
            if tag.startswith("hybrid") and name == "merged":  # (TieBrush-merged inputs carry tags the joined tile does not: the host loader takes them)
                assert "hybrid" not in r.stderr
            assert "device writer:" in r.stderr, (tag, r.stderr)   # (the default writer: tags + BGZF deflate on the GPU, devwriter.h)
            raw = bamio.bgzf_decompress(open(out, "rb").read())
            streams[tag] = raw[bamio.parse_header(raw)[1]:]
            # the same run through the host writer (every core tags and deflates): the same records byte for byte
            outh = str(tmp_path / ("%s_%s_hostwriter.bam" % (name, tag)))
            rh = subprocess.run([os.path.join(BIN, "tiebrush"), "--writer", "host", "-o", outh] + paths, check=True, capture_output=True, text=True,
                                env=dict(os.environ, TBK_TIMING="1", **env))
            assert "device writer:" not in rh.stderr
            rawh = bamio.bgzf_decompress(open(outh, "rb").read())
            assert rawh[bamio.parse_header(rawh)[1]:] == streams[tag], (name, tag)   # (the headers differ in the @PG line's command line)
        assert len(streams["whole"]) > 100000
        for tag in ("stream", "tiles", "device", "zlib", "whole_nomem", "device_nomem", "hybrid", "hybrid_80", "hybrid_nomem"):
            assert streams[tag] == streams["whole"], (name, tag)


def test_cli_streams_many_inputs_with_few_descriptors_and_small_tiles(tmp_path):
    """The streaming driver (TInputFiles::next_tile): 300 inputs under `ulimit -n 64` (an input's descriptor is only open
    while its window is refilled), tiles of ~2000 records cut at global bundle boundaries — the output equals the oracle's
    flat collapse of the same inputs record for record, YC / YX / YD included, and equals the one-tile run byte for byte."""
    import resource
    from oracle import oracle_ffi as orc
    from tiebrush_amd import bamio, synth
    tile = synth.make_tile(300, 120, "c5", n_loci=300)
    paths = synth.write_bams(tile, str(tmp_path / "m"))
    lst = str(tmp_path / "inputs.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    flat = orc.collapse(tile, strategy=3, max_nh=5, min_qual=1)
    args = ["-E", "-N", "5", "-Q", "1"]

    def limit():
        resource.setrlimit(resource.RLIMIT_NOFILE, (64, 64))

    one, tiled = str(tmp_path / "one.bam"), str(tmp_path / "tiled.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", one] + args + [lst])
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "-o", tiled] + args + [lst], check=True, capture_output=True, text=True,
                       env=dict(os.environ, TBK_TILE_RECORDS="2000", TBK_TIMING="1", TBK_DEVICE_DECODE="0"), preexec_fn=limit)
    assert "%d input records written as %d" % (flat["n_passed"], flat["n_groups"]) in r.stderr
    ntiles = int([l for l in r.stderr.splitlines() if l.startswith("tiles:")][0].split()[1])
    assert ntiles > 5
    ra, rb = bamio.bgzf_decompress(open(one, "rb").read()), bamio.bgzf_decompress(open(tiled, "rb").read())
    assert ra[bamio.parse_header(ra)[1]:] == rb[bamio.parse_header(rb)[1]:]
    out = bamio.read_bam(tiled)
    assert out.n == flat["n_groups"]
    rep = flat["rep"].astype(np.int64)
    assert np.array_equal(out.pos, tile.pos[rep]) and np.array_equal(out.tid, tile.tid[rep])
    assert np.array_equal(out.yc.astype(np.float64), flat["yc"].astype(np.float32).astype(np.float64))
    assert np.array_equal(out.yx, flat["yx"]) and np.array_equal(out.yd, flat["yd"])


def test_tiecov_bigwig_output(tmp_path):
    """tiecov -W (tiecov.cpp:243-275, :365-402): the coverage goes to PREFIX.bigwig; read back with the tests' own bigWig
    parser it holds the intervals of the golden bedGraph (values as float32), junctions still go to the BED file"""
    from bigwig_reader import BigWig
    pre = str(tmp_path / "t1")
    _run([os.path.join(BIN, "tiecov"), "-W", "-c", pre + ".coverage", "-j", pre + ".junctions", os.path.join(GOLDEN, "t1", "t1.bam")])
    assert not os.path.exists(pre + ".coverage.bedgraph")
    want = []
    for ln in read_lines(os.path.join(GOLDEN, "t1", "t1.coverage.bedgraph"))[1:]:
        c, a, b, v = ln.split("\t")
        want.append((c, int(a), int(b), float(np.float32(float(v)))))
    bw = BigWig(pre + ".coverage.bigwig")
    assert bw.intervals() == want
    assert bw.summary[0] == sum(b - a for _, a, b, _ in want) and bw.n_zoom >= 1
    assert len(read_lines(pre + ".junctions.bed")) == len(read_lines(os.path.join(GOLDEN, "t1", "t1.junctions.bed")))


def test_sam_text_inputs(tmp_path):
    """SAM inputs (GSam.h:371-401 opens them like BAM): the ten t2 samples as SAM text give the golden t2.bam, the collapsed
    file as SAM text gives the golden tracks; a single SAM argument is an input, not a list of paths"""
    from samtext import bam_to_sam_text
    sams = []
    for i, p in enumerate(sample_paths("t2")):
        s = tmp_path / ("t2s%d.sam" % i)
        s.write_text(bam_to_sam_text(p)[0])
        sams.append(str(s))
    out = str(tmp_path / "o.bam")
    _run([os.path.join(BIN, "tiebrush"), "-A", "-o", out] + sams)
    _compare_bam(out, os.path.join(GOLDEN, "t2", "t2.bam"))
    one = str(tmp_path / "one.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", one, sams[0]])
    from tiebrush_amd import bamio
    assert bamio.read_bam(one).n > 0
    csam = tmp_path / "t2.sam"
    csam.write_text(bam_to_sam_text(os.path.join(GOLDEN, "t2", "t2.bam"))[0])
    pre = str(tmp_path / "t2")
    _run([os.path.join(BIN, "tiecov"), "-c", pre + ".coverage", str(csam)])
    got = [l[:-4] if l.endswith(".000") else l for l in read_lines(pre + ".coverage.bedgraph")]
    assert got == read_lines(os.path.join(GOLDEN, "t2", "t2.coverage.bedgraph"))


def test_tiles_are_not_cut_behind_an_intron_ending_cigar(tmp_path):
    """the streaming host may cut a tile only where the next read starts beyond end + 1 of everything before it: the YD
    lists can hold a node (end + 1, end) (a CIGAR that ends in an intron); with two-record tiles the output still equals the
    one-tile output and the oracle"""
    from oracle import oracle_ffi as orc
    from test_gpu_collapse import _degenerate_exon_files
    from test_gpu_window import _tile
    from tiebrush_amd import bamio, synth
    tile = _tile(_degenerate_exon_files())
    paths = synth.write_bams(tile, str(tmp_path / "in"))
    want = orc.collapse(tile)
    outs = []
    for tr in ("2", "1000000"):
        out = str(tmp_path / ("o%s.bam" % tr))
        _run([os.path.join(BIN, "tiebrush"), "-o", out] + paths, env=dict(os.environ, TBK_TILE_RECORDS=tr, TBK_DEVICE_DECODE="0"))
        o = bamio.read_bam(out)
        assert o.n == want["n_groups"]
        assert [int(x) for x in o.yd] == [int(x) for x in want["yd"]] and [int(x) for x in o.yx] == [int(x) for x in want["yx"]]
        outs.append(open(out, "rb").read())


@pytest.mark.parametrize("env", [dict(TBK_TILE_RECORDS="3"), dict(TBK_HOST_FAST="0"), dict(), dict(TBK_DEVICE_DECODE="1"), dict(TBK_NO_LIBDEFLATE="1")],
                         ids=["host-tiny-tiles", "host-streaming-one-tile", "host-whole-input", "device-decode", "whole-input-zlib"])
def test_real_bam_shapes_through_the_command_line(tmp_path, env):
    """unmapped mates in place, an unplaced tail, an input without records, a CIGAR ending in an intron: every decode path of
    the command line gives the oracle's records, in its order, with its tags"""
    from helpers import paired_end_like_files, tile_from_records
    from oracle import oracle_ffi as orc
    from tiebrush_amd import bamio, synth
    tile = tile_from_records(paired_end_like_files())
    paths = synth.write_bams(tile, str(tmp_path / "in"))
    want = orc.collapse(tile)
    out = str(tmp_path / "o.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", out] + paths, env=dict(os.environ, **env))
    o = bamio.read_bam(out)
    fo = tile.file_of()
    assert o.n == want["n_groups"]
    assert [bytes(x) for x in o.qname] == [b"r%d_%d" % (int(fo[g]), int(g) - int(tile.file_off[int(fo[g])])) for g in want["rep"]]
    assert [int(x) for x in o.yd] == [int(x) for x in want["yd"]] and [int(x) for x in o.yx] == [int(x) for x in want["yx"]]
    assert [float(x) for x in o.yc] == [float(np.float32(x)) for x in want["yc"]]


@pytest.mark.parametrize("name,flags", [("t1", []), ("t2", ["-P"])])
def test_tiebrush_ranks_two_processes_end_in_one_bam(tmp_path, name, flags):
    """`tiebrush --ranks 2`: two processes (one GPU shared through the gloo staging hook; RCCL needs a GPU per rank), five sample files
    each — local collapse, group partials exchanged and reduced by range owner, the winners fetched back from the rank that holds
    the file, tagged and deflated per rank, one BAM.  Its records equal the single-GPU command line's byte for byte, and the
    golden BAM through the normaliser (HEAD without -A differs from the 0.0.6 goldens by YC + 1 at the records SURVEY.md §4.4
    lists)."""
    from tiebrush_amd import bamio
    out1, out2 = str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    ins = sample_paths(name)
    _run([os.path.join(BIN, "tiebrush"), "-o", out1] + flags + ins)
    env = dict(os.environ, TBK_RANKS_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "--ranks", "2", "-o", out2] + flags + ins, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = bamio.read_bam(out1, keep_aux=True), bamio.read_bam(out2, keep_aux=True)
    assert a.n == b.n
    for i in range(a.n):
        assert bamio.record_bytes(a, i) == bamio.record_bytes(b, i), i          # every byte of every record, tags included
    n_in = {"t1": 416922, "t2": 242910}[name]
    assert "%d input records written as %d" % (n_in, b.n) in r.stderr
    assert b.header.is_tiebrush() and len(b.header.co_samples()) == 10 and not os.path.exists(out2 + ".part0")
    g = bamio.read_bam(os.path.join(GOLDEN, name, name + ".bam"))
    deltas = {"t1": [1930, 2210], "t2": [2233, 4901, 5655, 8154]}[name]
    assert g.n == b.n
    for i in range(g.n):
        assert bamio.record_identity(b, i) == bamio.record_identity(g, i), i
        gyc = (g.yc[i] if g.has_yc[i] else 1.0) + (1.0 if i in deltas else 0.0)
        assert b.yc[i] == gyc and b.yx[i] == g.yx[i] and b.yd[i] == g.yd[i], i


def test_tiebrush_ranks_recollapse_of_tiebrush_outputs(tmp_path):
    """`tiebrush --ranks 2` on the two golden tissue BAMs (TieBrush-merged inputs: carried YC / YX / YD, records that already hold
    the tags and take htslib's in-place update rules) == golden t12.bam == the single-GPU command line, byte for byte"""
    from tiebrush_amd import bamio
    ins = [os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")]
    out1, out2 = str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", out1] + ins)
    env = dict(os.environ, TBK_RANKS_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "--ranks", "2", "-o", out2] + ins, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = bamio.read_bam(out1, keep_aux=True), bamio.read_bam(out2, keep_aux=True)
    assert a.n == b.n == 9491
    for i in range(a.n):
        assert bamio.record_bytes(a, i) == bamio.record_bytes(b, i), i
    g = bamio.read_bam(os.path.join(GOLDEN, "t12.bam"))
    for i in range(g.n):     # (the inputs carry 0.0.6's integer YC: htslib's float update fails on it and the stale value survives, as in
        assert bamio.record_identity(b, i) == bamio.record_identity(g, i)            # test_tiebrush_cli_recollapse_and_listfile)
        assert b.yx[i] == g.yx[i] and b.yd[i] == g.yd[i]


def test_tiebrush_ranks_three_ranks_uneven_files(tmp_path):
    """ten files over three ranks (3 + 3 + 4), --exon: same records as one GPU"""
    from tiebrush_amd import bamio
    ins = sample_paths("t2")
    out1, out2 = str(tmp_path / "one.bam"), str(tmp_path / "three.bam")
    _run([os.path.join(BIN, "tiebrush"), "-E", "-o", out1] + ins)
    env = dict(os.environ, TBK_RANKS_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "--ranks", "3", "-E", "-o", out2] + ins, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = bamio.read_bam(out1, keep_aux=True), bamio.read_bam(out2, keep_aux=True)
    assert a.n == b.n
    for i in range(a.n):
        assert bamio.record_bytes(a, i) == bamio.record_bytes(b, i), i


def _records(path):
    from tiebrush_amd import bamio
    raw = bamio.bgzf_decompress(open(path, "rb").read())
    return raw[bamio.parse_header(raw)[1]:]


def _members_ok(path):
    """every BGZF member of the file: framing, CRC32, ISIZE, zlib inflates it; the last one is the EOF member"""
    import struct
    import zlib
    run = open(path, "rb").read()
    o, sizes = 0, []
    while o < len(run):
        assert run[o:o + 4] == b"\x1f\x8b\x08\x04" and run[o + 12:o + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", run, o + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", run, o + bsize - 8)
        d = zlib.decompressobj(-15)
        got = d.decompress(run[o + 18:o + bsize - 8]) + d.flush()
        assert d.eof and len(got) == isize and zlib.crc32(got) & 0xFFFFFFFF == crc
        sizes.append(isize)
        o += bsize
    assert sizes[-1] == 0 and run[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return sizes


@pytest.mark.parametrize("case", ["t1", "t2", "t12", "t2_clip", "t2_exon"])
def test_device_writer_equals_host_writer_on_goldens(tmp_path, case):
    """the output side on the GPU (tbk_bam_encode) against the host writer: same record stream byte for byte, well-formed members,
    compressed size within 10 % of the host's zlib / libdeflate level 6"""
    paths = {"t1": sample_paths("t1"), "t2": sample_paths("t2"), "t12": [os.path.join(GOLDEN, "t1", "t1.bam"), os.path.join(GOLDEN, "t2", "t2.bam")],
             "t2_clip": sample_paths("t2"), "t2_exon": sample_paths("t2")}[case]
    flags = {"t2_clip": ["-P"], "t2_exon": ["-E"]}.get(case, [])
    d, h = str(tmp_path / "dev.bam"), str(tmp_path / "host.bam")
    rd = subprocess.run([os.path.join(BIN, "tiebrush"), "-o", d] + flags + paths, check=True, capture_output=True, text=True, env=dict(os.environ, TBK_TIMING="1"))
    rh = subprocess.run([os.path.join(BIN, "tiebrush"), "--writer", "host", "-o", h] + flags + paths, check=True, capture_output=True, text=True,
                        env=dict(os.environ, TBK_TIMING="1"))
    assert "device writer:" in rd.stderr and "device writer:" not in rh.stderr
    assert _records(d) == _records(h)
    _members_ok(d)
    sd, sh = os.path.getsize(d), os.path.getsize(h)
    print("%s: %d bytes from the device writer, %d from the host writer: %.3f" % (case, sd, sh, sd / sh))
    assert sd <= 1.10 * sh
    # (the goldens themselves: test_tiebrush_cli_on_samples and the t12 test above run through this writer, the default)


@pytest.mark.parametrize("profile,flags", [("c5", ["-E", "-N", "5", "-Q", "1"]), ("c3", ["--clip"]), ("c5", ["--keep-secondary", "-S", "--store-frac"]),
                                           ("c2", []), ("c2", ["-A"]), ("c2", ["-L"])])
def test_device_writer_equals_host_writer_on_the_option_matrix(tmp_path, profile, flags):
    from tiebrush_amd import synth
    tile = synth.make_tile(4, 6000, profile, n_loci=80)
    paths = synth.write_bams(tile, str(tmp_path / "in"))
    d, h = str(tmp_path / "dev.bam"), str(tmp_path / "host.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", d] + flags + paths)
    _run([os.path.join(BIN, "tiebrush"), "--writer", "host", "-o", h] + flags + paths)
    assert _records(d) == _records(h)
    _members_ok(d)


def test_writer_option_is_checked():
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "--writer", "tape", "-o", "/tmp/x.bam", os.path.join(GOLDEN, "t12.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and "--writer takes host or device" in r.stderr


def test_tiebrush_ranks_full_strategy_equals_single_gpu(tmp_path):
    """`tiebrush --ranks 2 -L`: grouping by CIGAR and MD across ranks — the representatives' MD strings travel beside the partial rows
    (tbk_partial_pack_md) and the owner compares them like cmpFull (tiebrush.cpp:285-302).  Four inputs whose records carry MD strings
    that differ inside a CIGAR group (and records without the tag), two per rank: the records equal the single-GPU `tiebrush -L` byte
    for byte, -L groups differently from the default; and the four t2 samples of the reference the same way"""
    from tiebrush_amd import bamio
    rng = np.random.default_rng(43)
    variants = [b"MDZ100\0", b"MDZ50A49\0", b"MDZ10^AC90\0", b"MDZ\0", b""]
    hdr = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n"
    ins = []
    for f in range(4):
        pos = np.sort(rng.integers(100, 400, 3000))
        recs = b"".join(bamio.encode_record(0, int(p), 16 * int(rng.integers(0, 2)), 60, [100 << 4] if rng.integers(0, 4) else [40 << 4, (200 << 4) | 3, 60 << 4],
                                            b"r%d_%d" % (f, i), aux=b"NHC\x01" + variants[int(rng.integers(0, 5))], l_seq=0) for i, p in enumerate(pos))
        path = str(tmp_path / ("md%d.bam" % f))
        bamio.write_bam(path, hdr, ["chr1"], [100000], recs)
        ins.append(path)
    env = dict(os.environ, TBK_RANKS_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for tag, inputs in (("synthetic", ins), ("t2", sample_paths("t2")[:4])):
        one, two, dflt = str(tmp_path / (tag + "_one.bam")), str(tmp_path / (tag + "_two.bam")), str(tmp_path / (tag + "_default.bam"))
        _run([os.path.join(BIN, "tiebrush"), "-L", "-o", one] + inputs)
        _run([os.path.join(BIN, "tiebrush"), "-o", dflt] + inputs)
        r = subprocess.run([os.path.join(BIN, "tiebrush"), "--ranks", "2", "-L", "-o", two] + inputs, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        a, b, c = bamio.read_bam(one, keep_aux=True), bamio.read_bam(two, keep_aux=True), bamio.read_bam(dflt)
        assert a.n == b.n and (a.n > c.n if tag == "synthetic" else a.n >= c.n)      # (MD splits groups the CIGAR alone joins)
        for i in range(a.n):
            assert bamio.record_bytes(a, i) == bamio.record_bytes(b, i), (tag, i)


def test_tiled_run_writes_behind_the_input_side(tmp_path):
    """the bounded-memory path on records with SEQ / QUAL: several tiles, each written by the writer thread (its own context) while the
    next one is inflated and collapsed — the same record stream as the single-tile run, with either writer, well-formed members"""
    from tiebrush_amd import synth
    tile = synth.make_tile(4, 40000, "c2", n_loci=300)
    paths = synth.write_bams_fast(tile, str(tmp_path / "in"), seq=True)
    whole = str(tmp_path / "whole.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", whole] + paths)
    for writer in ("device", "host"):
        out = str(tmp_path / ("tiled_%s.bam" % writer))
        r = subprocess.run([os.path.join(BIN, "tiebrush"), "--writer", writer, "-o", out] + paths, check=True, capture_output=True, text=True,
                           env=dict(os.environ, TBK_TILE_RECORDS="30000", TBK_TIMING="1"))
        assert "streamed:" in r.stderr and "writer thread busy" in r.stderr, r.stderr
        ntiles = int(r.stderr.split("streamed:")[1].split("tiles")[0])
        assert ntiles >= 3, r.stderr
        assert ("device writer:" in r.stderr) == (writer == "device")
        assert _records(out) == _records(whole), writer
        _members_ok(out)


def _long_record_input(tmp_path, n_before=700, n_after=150, long_len=70000):
    """one BAM: n_before distinct 50-bp reads, then ONE read of long_len bases (a record of ~ 105 KB: longer than a BGZF member's 64 KB
    payload, what tbk_bam_encode refuses), then n_after more — every read a group of its own, in coordinate order"""
    from tiebrush_amd import bamio
    recs = []
    for i in range(n_before):
        recs.append(bamio.encode_record(0, 100 + 10 * i, 0, 60, [(50 << 4) | 0], b"s%d" % i, aux=b"NHC\x01", l_seq=50, seq=bytes(25), qual=bytes([30]) * 50))
    p0 = 100 + 10 * n_before
    recs.append(bamio.encode_record(0, p0, 0, 60, [(long_len << 4) | 0], b"long", aux=b"NHC\x01", l_seq=long_len, seq=bytes((long_len + 1) // 2),
                                    qual=bytes([30]) * long_len))
    for i in range(n_after):
        recs.append(bamio.encode_record(0, p0 + 10 + 10 * i, 0, 60, [(50 << 4) | 0], b"t%d" % i, aux=b"NHC\x01", l_seq=50, seq=bytes(25), qual=bytes([30]) * 50))
    path = str(tmp_path / "long.bam")
    bamio.write_bam(path, "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:50000000\n", ["chr1"], [50000000], b"".join(recs))
    return path, n_before + 1 + n_after


@pytest.mark.parametrize("mode", ["long-record", "forced"])
def test_device_writer_refusal_on_a_later_chunk_hands_over_to_the_host_writer(tmp_path, mode):
    """A chunk the device writer cannot take (a record longer than a BGZF member — a long read — or no memory) used to end the run when it
    was not the first one; now the chunks before it stay written and the host writer goes on from that group: the record stream equals
    the host writer's, whatever chunk refuses.  TBK_DW_CHUNK_GROUPS makes a small input span several chunks."""
    path, n = _long_record_input(tmp_path)
    env = dict(os.environ, TBK_TIMING="1", TBK_DW_CHUNK_GROUPS="256")
    if mode == "forced":
        env["TBK_TEST_DW_REFUSE_CHUNK"] = "1"
        path = os.path.join(GOLDEN, "t12.bam")
    d, h = str(tmp_path / "dev.bam"), str(tmp_path / "host.bam")
    rd = subprocess.run([os.path.join(BIN, "tiebrush"), "-o", d, path], check=True, capture_output=True, text=True, env=env)
    subprocess.run([os.path.join(BIN, "tiebrush"), "--writer", "host", "-o", h, path], check=True, capture_output=True, text=True)
    assert _records(d) == _records(h)
    want_at = {"long-record": 512, "forced": 256}[mode]      # (the long read is group 700: chunk 2 of 256-group chunks)
    assert "device writer stopped after %d of" % want_at in rd.stderr, rd.stderr[-600:]
    _members_ok(d)
    if mode == "long-record":
        from tiebrush_amd import bamio
        assert bamio.read_bam(d).n == n


def test_tiebrush_ranks_four_ranks_equal_single_gpu(tmp_path):
    """ten files over four ranks (3 + 3 + 2 + 2): the multi-rank command line ends in the single-GPU run's BAM, record for record (four
    processes share the box's GPU through the gloo hook: what its process guard leaves beside the test runner)"""
    from tiebrush_amd import bamio
    ins = sample_paths("t1")
    out1, out2 = str(tmp_path / "one.bam"), str(tmp_path / "four.bam")
    _run([os.path.join(BIN, "tiebrush"), "-o", out1] + ins)
    env = dict(os.environ, TBK_RANKS_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(BIN, "tiebrush"), "--ranks", "4", "-o", out2] + ins, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = bamio.read_bam(out1, keep_aux=True), bamio.read_bam(out2, keep_aux=True)
    assert a.n == b.n == 3479
    for i in range(a.n):
        assert bamio.record_bytes(a, i) == bamio.record_bytes(b, i), i
