"""describe_kernel (moped_amd/csrc/sift.hip) walks only the columns of a descriptor window's rows that
`desc_row_interval` keeps; the interval has to contain every sample that passes KeySample's tests (libsiftfast
MakeKeypointSample / KeySample, libs.tgz -> libsiftfast.cpp:1560-1567) -- checked here on the host, the same header
compiled by g++ without contraction against the tests themselves, over random keys (all sizes the shipped constants
produce, orientations on and next to the axes where a condition's slope vanishes, windows cut by the image border)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_row_intervals_contain_every_passing_sample(tmp_path):
    exe = str(tmp_path / "sift_rows_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HERE, "native", "sift_rows_check.cpp"), "-o", exe],
                   check=True)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    last = out.stdout.strip().splitlines()[-1].split()
    assert last[0] == "OK"
    overhead = float(last[-1])
    assert 1.0 <= overhead < 1.25, out.stdout   # the walk visits at most a quarter more samples than pass
