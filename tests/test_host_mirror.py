"""Host-side mirror of the reference's .sfs text format (ping_pong.cpp:224-230, sfs.cpp:5-30)."""
import numpy as np

import svdss_amd


def test_output_batch_and_parse_roundtrip():
    sols = [("read/1", 0, [(10, 5), (40, 7)]), ("read/2", 2, []), ("read/3", 1, [(0, 3)])]
    txt = svdss_amd.output_batch(sols)
    assert txt == "read/1\t10\t5\t0\t\n*\t40\t7\t0\t\nread/3\t0\t3\t1\t\n"
    back = svdss_amd.parse_sfsfile(txt)
    assert back == {"read/1": [(10, 5, 0), (40, 7, 0)], "read/3": [(0, 3, 1)]}


def test_pack_reads():
    flat, offs = svdss_amd.pack_reads(["ACGT", np.array([1, 5], np.uint8), ""])
    assert flat.tolist() == [1, 2, 3, 4, 1, 5] and offs.tolist() == [0, 4, 6, 6]
    flat, offs = svdss_amd.pack_reads([])
    assert len(flat) == 0 and offs.tolist() == [0]
