"""The dense loops on hosts that do not behave: the resident fit launch and the dense kernels talk through doorbells
across streams, forks and joins ride on dispatch completion signals, and the library asks for more hardware queues when it is loaded -
all of which assume things about the process around them.  Here the process initialises HIP first (the queue request comes too late,
or the 'user' has pinned GPU_MAX_HW_QUEUES to 2), or hammers the device from two other streams while the loops run.  Whatever schedule
the self-tests then choose, the results must be the bits of an undisturbed run - and a fit must never hang or fail."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def undisturbed():
    import hostile_child
    return hostile_child.jobs()


def _child(tmp_path, target, *args):
    import conftest
    if conftest.FRESH is None:
        pytest.skip("no fork server")
    out = str(tmp_path / "out.npz")
    p = conftest.FRESH.Process(target=target, args=(out,) + args)
    p.start()
    p.join(600)
    if p.is_alive():
        p.terminate()
        pytest.fail("the child hung")
    err = tmp_path / "out.npz.err"
    assert p.exitcode == 0, "exit code %s\n%s" % (p.exitcode, err.read_text() if err.exists() else "")
    return np.load(out)


def _same(got, want):
    for key in ("mask_params", "scan_params", "scan_disp"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


@pytest.mark.parametrize("max_queues", [0, 2])
def test_hip_initialised_before_the_library_is_loaded(tmp_path, undisturbed, max_queues):
    import hostile_child
    got = _child(tmp_path, hostile_child.hip_first, max_queues)
    print("HIP first, GPU_MAX_HW_QUEUES=%r: resident fit launch in the silhouette loop %d, in the scan loop %d (undisturbed: %d, %d)" % (
        str(got["queues_env"]), int(got["mask_resident"]), int(got["scan_resident"]), int(undisturbed["mask_resident"]), int(undisturbed["scan_resident"])))
    _same(got, undisturbed)


def test_other_streams_keep_the_device_busy(tmp_path, undisturbed):
    import hostile_child
    got = _child(tmp_path, hostile_child.busy_neighbour)
    print("busy neighbour: %d rounds of 2 x 256 MB memsets beside the loops; resident fit launch %d / %d" % (
        int(got["neighbour_rounds"]), int(got["mask_resident"]), int(got["scan_resident"])))
    assert int(got["neighbour_rounds"]) > 0
    _same(got, undisturbed)
