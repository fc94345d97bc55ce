"""MATLAB -v7.3 (HDF5) dictionary files, se_snmf_nat_amd/mat73.py: run_basis_train.m:136 writes R_<R>.mat with `-v7.3` and :138
loads it back; the three files the reference SHIPS are MAT-5, which is why nothing noticed until round 3's review that a
dictionary trained by the reference's own MATLAB could not be read.

Fixtures (tests/golden/, made by the snippet in this file's docstring of `test_fixture_written_by_libhdf5`):
  basis_v73_small.mat             written by mat73.save_mat73 (contiguous float64, symbol-table group, 512-byte MATLAB user block)
  basis_v73_deflate_libhdf5.mat   the same variables re-written by the HDF5 LIBRARY's h5repack 1.10.6 as chunked datasets with the
                                  shuffle + deflate filters (`save -v7.3` compresses by default): what the reader must handle
  basis_v73_small_expected.npz    the arrays
Where the HDF5 tools are installed (/opt/conda/bin/h5dump) the writer's output is also read back by the library itself.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from se_snmf_nat_amd import mat73

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def _expected():
    return dict(np.load(os.path.join(GOLD, "basis_v73_small_expected.npz")))


def test_round_trip_and_layout(tmp_path):
    rs = np.random.RandomState(1)
    d = {"B_DFT_sub": rs.rand(513, 100), "B_Mel_sub": rs.rand(64, 100), "A_DFT_sub": 0.0, "A_Mel_sub": rs.rand(100, 7), "v": np.arange(5.0)}
    f = str(tmp_path / "R_100.mat")
    mat73.save_mat73(f, d)
    raw = open(f, "rb").read()
    # the published layout: 512-byte user block that starts with MATLAB's text, version 0x0200 + "IM" at 124, HDF5 signature at 512
    assert raw[:19] == b"MATLAB 7.3 MAT-file" and raw[124:128] == b"\x00\x02IM" and raw[512:520] == b"\x89HDF\r\n\x1a\n"
    assert mat73.is_mat73(f)
    back = mat73.load_mat73(f)
    assert set(back) == set(d)
    for k in d:
        assert np.array_equal(back[k], np.atleast_2d(np.asarray(d[k], dtype=np.float64))), k  # bits, MATLAB shapes (scalars 1 x 1, vectors 1 x n)
    with pytest.raises(ValueError):
        mat73.save_mat73(f, {"not a name": 1.0})
    with pytest.raises(ValueError):
        mat73.load_mat73(os.path.join(GOLD, "basis_v73_small_expected.npz"))  # not an HDF5 file


def test_fixture_written_by_the_writer():
    got = mat73.load_mat73(os.path.join(GOLD, "basis_v73_small.mat"))
    exp = _expected()
    assert set(got) == set(exp) and all(np.array_equal(got[k], exp[k]) for k in exp)


def test_fixture_written_by_libhdf5():
    """Chunked + shuffle + deflate datasets indexed by v1 B-trees, written by the HDF5 library itself:
        h5repack -f SHUF -f GZIP=3 -l A_DFT_sub:CHUNK=8x3 -l A_Mel_sub:CHUNK=21x4 -l B_DFT_sub:CHUNK=2x9 -l B_Mel_sub:CHUNK=4x5 \\
                 -u <first 512 bytes of basis_v73_small.mat> -b 512 basis_v73_small.mat basis_v73_deflate_libhdf5.mat
    (edge chunks are partial: 21 = 2 * 8 + 5 rows, 4 = 3 + 1 columns)."""
    got = mat73.load_mat73(os.path.join(GOLD, "basis_v73_deflate_libhdf5.mat"))
    exp = _expected()
    assert set(got) == set(exp) and all(np.array_equal(got[k], exp[k]) for k in exp)


def test_training_driver_reads_and_writes_v73(tmp_path):
    """train.save_basis_mat / load_basis_mat: -v7.3 as run_basis_train.m:136 writes it, MAT-5 as the shipped files are -- `load` takes both."""
    from se_snmf_nat_amd import train
    exp = _expected()
    for v73 in (True, False):
        f = str(tmp_path / ("a.mat" if v73 else "b.mat"))
        train.save_basis_mat(f, exp, v73=v73)
        assert mat73.is_mat73(f) == v73
        back = train.load_basis_mat(f)
        assert set(back) == set(exp) and all(np.array_equal(back[k], exp[k]) for k in exp)
    back = train.load_basis_mat(os.path.join(GOLD, "basis_v73_deflate_libhdf5.mat"))
    np.testing.assert_allclose(np.sqrt(((back["B_DFT_sub"] - 1e-9) ** 2).sum(0)), 1.0, atol=1e-12)  # run_basis_train.m:113-114


@pytest.mark.skipif(H5DUMP is None, reason="HDF5 tools not installed")
def test_the_hdf5_library_reads_what_the_writer_wrote(tmp_path):
    rs = np.random.RandomState(2)
    d = {"B_DFT_sub": rs.rand(6, 3), "A_Mel_sub": rs.rand(3, 11)}
    f = str(tmp_path / "w.mat")
    mat73.save_mat73(f, d)
    out = subprocess.run([H5DUMP, "-p", f], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout
    txt = out.stdout
    assert 'DATASET "B_DFT_sub"' in txt and "( 3, 6 )" in txt and "( 11, 3 )" in txt  # MATLAB m x n <-> HDF5 (n, m)
    assert txt.count('"double"') == 2 and "H5T_IEEE_F64LE" in txt and "CONTIGUOUS" in txt
    # every value, through the library's own reader
    one = subprocess.run([H5DUMP, "-d", "/B_DFT_sub", "-y", "-w", "0", f], stdout=subprocess.PIPE, text=True).stdout
    vals = [float(x) for x in one[one.index("DATA {") + 6:one.index("}", one.index("DATA {"))].replace(",", " ").split()]
    np.testing.assert_allclose(np.array(vals).reshape(3, 6).T, d["B_DFT_sub"], rtol=1e-5)  # (h5dump prints six significant digits)
