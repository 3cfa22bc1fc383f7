"""GPU test of das4whales_amd.stream.FileStream: consecutive files processed as one
continuous record.  Parity targets: the band-pass of every file equals the oracle's bp_filt of the
CONCATENATED record; the correlogram's last lags continue into the next file."""
import numpy as np
import pytest
import torch

from oracle import d4w_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5
FS = 200.0


def rel(y, ref):
    return float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize("with_fk", [False, True])
def test_stream_equals_concatenated_record(with_fk):
    assert torch.cuda.is_available()
    import das4whales_amd as dw
    from das4whales_amd import stream
    rng = np.random.default_rng(31)
    nx, ns, nfiles = 48, 3000, 4
    rec = rng.standard_normal((nx, ns * nfiles)) + 0.3
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    lf = orc.gen_template_fincall(t, FS, 14.7, 21.8, 0.78)
    mask = np.ones((nx, ns)) if with_fk else None            # identity f-k mask: exercises the per-file f-k step
    st = stream.FileStream(FS, 14, 30, templates=[hf, lf], fk_mask=mask, halo=1024)
    results = []
    for i in range(nfiles):
        got = st.push(rec[:, i * ns:(i + 1) * ns])
        assert len(got) == (1 if i >= 2 else 0)             # file i is final when file i + 2 has arrived
        results += got
    results += st.flush()
    assert [r["index"] for r in results] == list(range(nfiles))
    F = orc.bp_filt(rec, FS, 14, 30)                         # the reference filter on the whole record
    taps = [dw.detect._normalised_support(hf), dw.detect._normalised_support(lf)]
    filt = {r["index"]: r["filtered"].cpu().numpy().astype(np.float64) for r in results}
    for r in results:
        i = r["index"]
        Fi = F[:, i * ns:(i + 1) * ns]
        e = rel(filt[i], Fi)
        assert e < TOL, ("band-pass", i, e)
        # the correlogram stage on ITS input (the stream's own filtered files, so that the band-pass error allowed
        # above does not pass through the 1 / max|x| normalisation into this check): 1e-5
        Gi = filt[i]
        m = Gi.mean(axis=1, keepdims=True)
        A = np.max(np.abs(Gi), axis=1, keepdims=True)
        for tp, c in zip(taps, r["correlograms"]):
            L = len(tp)
            seg = Gi if i + 1 not in filt else np.concatenate((Gi, filt[i + 1][:, :L - 1]), axis=1)
            xn = (seg - m) / A
            n = xn.shape[1]
            ref = np.stack([orc.shift_xcorr(xn[k], np.pad(tp, (0, n - L)))[:ns] for k in range(nx)])
            e = rel(c.cpu().numpy(), ref)
            assert e < TOL, ("correlogram", i, e)
        # the per-row maxima the correlator leaves in its epilogue for files that continue (what a detection threshold is
        # set from, scripts/main_mfdetect.py:82,95) are the maxima of the stored correlograms, and correlogram_max the block's
        if "row_max" in r:
            for c, rm in zip(r["correlograms"], r["row_max"]):
                assert torch.equal(rm, c.max(dim=1).values)
                assert dw.detect.correlogram_max(c, rm) == float(c.max()) == dw.detect.correlogram_max(c)
        else:
            raise AssertionError("every file of a stream carries the row maxima of its correlograms (round 5: the last one too)")
    # a stand-alone file (the reference's per-file run) differs from the stream at the file edges
    alone = dw.dsp.bp_filt(rec[:, ns:2 * ns], FS, 14, 30)
    assert rel(alone, F[:, ns:2 * ns]) > 1e-3


def test_stream_at_the_ooi_file_shape():
    """Three consecutive 11 020 x 12 000 files (BASELINE configs[4] geometry): the middle file's band-pass is the halo form
    (d4w_fir_fft_halo_f32, neighbours read in place), its correlograms continue into the third file
    (d4w_xcorr_fft_cont_f32).  Rows are independent: eight of them against the oracle on the concatenated record."""
    import das4whales_amd as dw
    from das4whales_amd import stream
    nx, ns = 11020, 12000
    gen = torch.Generator(device="cuda").manual_seed(77)
    files = [torch.randn((nx, ns), device="cuda", generator=gen) + 0.2 for _ in range(3)]
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    lf = orc.gen_template_fincall(t, FS, 14.7, 21.8, 0.78)
    st = stream.FileStream(FS, 14, 30, templates=[hf, lf], fk_mask=None, halo=1024)
    results = []
    for f in files:
        results += st.push(f)
    results += st.flush()
    rows = [0, 1, 2, 777, 5509, 5510, nx - 2, nx - 1]
    rec = np.concatenate([f[rows].cpu().numpy().astype(np.float64) for f in files], axis=1)
    F = orc.bp_filt(rec, FS, 14, 30)
    taps = [dw.detect._normalised_support(hf), dw.detect._normalised_support(lf)]
    mid = results[1]
    assert mid["index"] == 1
    got = mid["filtered"][rows].cpu().numpy().astype(np.float64)
    e = rel(got, F[:, ns:2 * ns])
    print("stream 11020x12000, middle file: band-pass vs the concatenated record %.3e" % e)
    assert e < TOL
    nxt = results[2]["filtered"][rows].cpu().numpy().astype(np.float64)
    m, A = got.mean(axis=1, keepdims=True), np.max(np.abs(got), axis=1, keepdims=True)
    for tp, c in zip(taps, mid["correlograms"]):
        L = len(tp)
        xn = (np.concatenate((got, nxt[:, :L - 1]), axis=1) - m) / A
        ref = np.stack([orc.shift_xcorr(xn[k], np.pad(tp, (0, xn.shape[1] - L)))[:ns] for k in range(len(rows))])
        e = rel(c[rows].cpu().numpy(), ref)
        print("stream 11020x12000, middle file: correlogram (continued) %.3e" % e)
        assert e < TOL


def test_pinned_ingest_double_buffered_upload():
    """data_handle.PinnedIngest (reference contract data_handle.py:181-230): files that start in pinned host memory reach
    the device through two buffers on a side stream and give the strain data_handle.load_das_data_array gives for the same
    raw matrix; the slots are recycled in order over more files than slots, from the ingest's own buffers and from a
    reader's pinned pool."""
    import das4whales_amd as dw
    from das4whales_amd import data_handle
    nx, ns = 300, 6000
    meta = {"scale_factor": 1.7e-2, "fs": 200.0, "dx": 2.04}
    rng = np.random.default_rng(11)
    files = [rng.integers(-30000, 30000, size=(nx, ns)).astype(np.int32) for _ in range(5)]
    ing = data_handle.PinnedIngest((nx, ns), np.int32, depth=2)
    np.copyto(ing.host_array(0), files[0])
    ing.upload(0)
    outs = []
    for i in range(len(files)):
        if i + 1 < len(files):
            s = (i + 1) % 2
            ing.wait_host(s)
            np.copyto(ing.host_array(s), files[i + 1])
            ing.upload(s)
        x, tx, dist = ing.strain(i % 2, [0, nx, 1], meta)
        outs.append(x.clone())
    torch.cuda.synchronize()
    for f, x in zip(files, outs):
        ref, _, _ = data_handle.load_das_data_array(f, [0, nx, 1], meta)
        assert torch.equal(x, ref)
    pool = [torch.from_numpy(f).pin_memory() for f in files]
    for i, p in enumerate(pool):
        ing.upload(i % 2, p)
        x, _, _ = ing.strain(i % 2, [0, nx, 2], meta)
        ref, _, _ = data_handle.load_das_data_array(files[i], [0, nx, 2], meta)
        assert torch.equal(x, ref)
    with pytest.raises(ValueError):
        ing.upload(0, torch.zeros((nx, ns), dtype=torch.int32))              # pageable: refused


def test_staged_host_transfers_match_plain_copies():
    """_device.upload_f32 / download (pinned, chunked, dtype conversion on the host threads) against plain copies, for the
    dtypes and layouts the NumPy-in / NumPy-out calls meet: float64 / float32 / int32, C and Fortran order, a strided view."""
    from das4whales_amd import _device as dev
    rng = np.random.default_rng(3)
    a = rng.standard_normal((3000, 6000))                                    # 144 MB as float64: the staged path
    for src in (a, a.astype(np.float32), np.asfortranarray(a), a[::2, ::3], a[::-1, ::-1], (a * 1000).astype(np.int32), a > 0):
        got = dev.upload_f32(src)
        assert got.dtype == torch.float32 and got.is_contiguous()
        assert torch.equal(got.cpu(), torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)))
    big = np.concatenate((a, a), axis=0)[::-1]                                # 288 MB, negative row stride: several chunks
    assert torch.equal(dev.upload_f32(big).cpu(), torch.from_numpy(np.ascontiguousarray(big, dtype=np.float32)))
    y = torch.randn((3000, 6000), device="cuda")
    for dt in (np.float64, np.float32):
        h = dev.download(y, dt)
        assert h.dtype == dt and np.array_equal(h, y.cpu().numpy().astype(dt))


@pytest.mark.parametrize("nx,ns,halo", [(47, 9000, 1024), (64, 6000, 600), (33, 9000, 256), (40, 3000, 1024)])
def test_edge_files_with_and_without_the_copies(nx, ns, halo, monkeypatch):
    """The first / last file of a record: the halo band-pass with a stand-in halo on the free side + that side's row-end
    pieces (round 5) against the concatenate / filter / crop form of rounds 3-4 (D4W_STREAM_EDGE_FIR=0), and both against the
    oracle on the concatenated record -- long and short files, a halo shorter than the filter's half width and rows too short
    for the overlap-save form (both fall back by themselves), an odd channel count."""
    import das4whales_amd as dw
    from das4whales_amd import stream
    rng = np.random.default_rng(nx + ns + halo)
    nfiles = 3
    rec = rng.standard_normal((nx, ns * nfiles)) + 0.2
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)
    F = orc.bp_filt(rec, FS, 14, 30)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("D4W_STREAM_EDGE_FIR", mode)
        st = stream.FileStream(FS, 14, 30, templates=[hf], fk_mask=None, halo=halo)
        res = []
        for i in range(nfiles):
            res += st.push(rec[:, i * ns:(i + 1) * ns])
        res += st.flush()
        outs[mode] = {r["index"]: (r["filtered"].cpu().numpy().astype(np.float64), r["correlograms"][0].cpu().numpy().astype(np.float64)) for r in res}
    for i in range(nfiles):
        Fi = F[:, i * ns:(i + 1) * ns]
        tol = TOL if halo >= 1024 else 2e-4                      # a short halo cuts the response: the stream's own limit, both forms alike
        assert rel(outs["1"][i][0], Fi) < tol and rel(outs["0"][i][0], Fi) < tol, (i, rel(outs["1"][i][0], Fi), rel(outs["0"][i][0], Fi))
        assert rel(outs["1"][i][0], outs["0"][i][0]) < 2 * tol
        assert rel(outs["1"][i][1], outs["0"][i][1]) < 2 * tol


def test_stream_with_two_recycled_input_buffers():
    """A caller that alternates TWO float32 CUDA buffers and pushes them directly (no copy): file i's buffer is overwritten
    with file i + 2 right after push(i + 1) -- the lifetime push() documents.  The stream keeps its own copy of the `halo`
    tail columns (a view of the pushed tensor would hand the band-pass of file i + 1 a left halo from file i + 2)."""
    from das4whales_amd import stream
    gen = torch.Generator(device="cuda").manual_seed(5)
    nx, ns, nfiles = 64, 6000, 5
    rec = torch.randn((nx, ns * nfiles), device="cuda", generator=gen) + 0.2
    t = np.arange(ns) / FS
    hf = orc.gen_template_fincall(t, FS, 17.8, 28.8, 0.68)

    def run(recycle):
        st = stream.FileStream(FS, 14, 30, templates=[hf], halo=1024)
        bufs = [torch.empty((nx, ns), device="cuda"), torch.empty((nx, ns), device="cuda")]
        out = []
        for i in range(nfiles):
            if recycle:
                b = bufs[i % 2]
                b.copy_(rec[:, i * ns:(i + 1) * ns])          # overwrites file i - 2, whose successor has been pushed
            else:
                b = rec[:, i * ns:(i + 1) * ns].clone()
            out += st.push(b)
        out += st.flush()
        return {r["index"]: (r["filtered"].clone(), r["correlograms"][0].clone()) for r in out}

    a, b = run(False), run(True)
    assert sorted(a) == sorted(b) == list(range(nfiles))
    for i in range(nfiles):
        assert torch.equal(a[i][0], b[i][0]), ("band-pass of file %d depends on the recycled buffer" % i)
        assert torch.equal(a[i][1], b[i][1]), ("correlogram of file %d depends on the recycled buffer" % i)
    F = orc.bp_filt(rec.double().cpu().numpy(), FS, 14, 30)
    for i in range(nfiles):
        assert rel(b[i][0].cpu().numpy(), F[:, i * ns:(i + 1) * ns]) < TOL
