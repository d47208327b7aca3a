"""Boundary (-m gpu): where the one-off cost of a tree is paid, and who may touch a shared tree when.

SURVEY 8b "Validate": the reference pays its one-off cost -- the kernel build -- in CLCaster::validate (CLCaster.cpp:157-206), and
its compute() (:224-228) costs a frame from the first call on.  Here the one-off cost is what the SVO kernels derive from the
tree (the dense table of its top, the empty boxes: 0.6 s for a 4096^3 terrain); vrc_validate / vrc_prepare build it, a frame only
builds what it finds missing.  The derived structures live with the TREE (vrc_tree), which several handles -- on several host
threads -- may hold: the tree's guard covers a frame from the moment it copies the pointers until its kernel is enqueued."""
import threading
import time

import numpy as np
import pytest

import voxel_raycaster_amd as vrc

pytestmark = pytest.mark.gpu


def _caster(sc, w, h, validate=True, tree_from=None, **settings):
    c = vrc.CLCaster()
    assert c.init(0)
    for name, v in dict(octree_dimensions=sc["dim"], using_octree=0, max_distance=3 * sc["dim"], hit_records=0, **settings).items():
        assert c.add_to_settings_buffer(name, name.upper(), v)
    assert c.assign_octree_from(tree_from) if tree_from is not None else c.assign_octree(sc["octree"]), c.last_error()
    assert (c.assign_camera(sc["cam_dir"], sc["cam_pos"]) and c.create_viewport(w, h) and c.assign_lights(sc["lights"])
            and c.create_texture_atlas(sc["atlas"], (16, 16))), c.last_error()
    if validate:
        assert c.validate(), c.last_error()
    return c


def _wall_ms(c):
    t0 = time.perf_counter()
    assert c.compute(), c.last_error()
    return (time.perf_counter() - t0) * 1e3


def test_validate_pays_the_one_off_cost_and_the_first_frame_costs_a_frame():
    """after validate() the table and the boxes exist (no frame has run), and the first compute() of the tree
    takes what a warm frame takes (before: + 0.64 s at depth 12, inside a call documented as one frame)."""
    import bench
    sc = bench.build_scene(12)
    w, h = 1920, 1080
    warm = _caster(sc, w, h)                                  # loads the kernel instance; its own tree, its own structures
    for _ in range(3):
        _wall_ms(warm)
    c = _caster(sc, w, h, validate=False)
    assert c.memory_usage2()["box_bytes"] == 0 and c.memory_usage2()["coarse_bytes"] == 0
    t0 = time.perf_counter()
    assert c.validate(), c.last_error()
    validate_s = time.perf_counter() - t0
    m = c.memory_usage2()
    assert m["coarse_bytes"] > 0 and m["box_bytes"] > 0 and m["note"] == "", m   # built, before any frame
    first = _wall_ms(c)
    later = min(_wall_ms(c) for _ in range(5))
    print(f"\nvalidate {validate_s * 1e3:.0f} ms (boxes {m['box_build_seconds'] * 1e3:.0f} ms on the device); first frame {first:.2f} ms, warm frame {later:.2f} ms")
    assert validate_s > 0.5 * m["box_build_seconds"]          # that is where the build went
    assert first <= 2.0 * later + 1.0, (first, later)         # (+1 ms: the frame's own first-use allocations -- counter partials, events)
    assert np.array_equal(c.read_image().view(np.uint32), warm.read_image().view(np.uint32))
    # a second validate() finds everything in place
    t0 = time.perf_counter()
    assert c.validate()
    assert time.perf_counter() - t0 < 0.05 and c.memory_usage2()["box_build_seconds"] == m["box_build_seconds"]


def test_prepare_follows_settings_changed_after_validate_and_the_lazy_build_remains():
    import bench
    sc = bench.build_scene(10)
    c = _caster(sc, 640, 360)
    lc = c.memory_usage2()["coarse_log2"]
    assert lc >= 1
    assert c.compute()
    ref = c.read_image().view(np.uint32).copy()
    # another table level asked for after validate(): prepare() builds it ...
    assert c.add_to_settings_buffer("coarse_log2", "COARSE_LOG2", lc - 1) and c.prepare(), c.last_error()
    m = c.memory_usage2()
    assert m["coarse_log2"] == lc - 1 and m["box_bytes"] > 0
    assert c.compute() and np.array_equal(c.read_image().view(np.uint32), ref)
    # ... and without prepare() the frame that needs it does (the fallback)
    assert c.overwrite_setting("coarse_log2", lc) and c.compute(), c.last_error()
    assert c.memory_usage2()["coarse_log2"] == lc and np.array_equal(c.read_image().view(np.uint32), ref)
    # the array branch derives nothing; a handle without an octree has nothing to prepare
    e = vrc.CLCaster()
    assert e.init(0) and not e.prepare() and "octree" in e.last_error()


def test_a_level_that_cannot_be_had_is_retried_when_the_host_asks_again():
    """a failed build switched the structure off for the life of the tree.  Now the failure is remembered for the
    (level, root, depth) that failed, another level is tried at once, and setting coarse_log2 / empty_boxes again clears it."""
    import bench
    sc = bench.build_scene(10)
    c = _caster(sc, 320, 200, validate=False)
    assert c.validate(), c.last_error()
    assert c.compute()
    ref = c.read_image().view(np.uint32).copy()
    m = c.memory_usage2()
    assert m["note"] == "" and m["coarse_bytes"] > 0
    # (a real out-of-memory cannot be provoked safely on a shared box; the bookkeeping is exercised through its public face:
    # switching a structure off and on again through the settings must bring it back with an empty note)
    assert c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", 0) and c.compute() and not c.used_empty_boxes()
    assert c.overwrite_setting("empty_boxes", -1) and c.compute() and c.used_empty_boxes()
    assert c.memory_usage2()["note"] == "" and np.array_equal(c.read_image().view(np.uint32), ref)


def test_two_host_threads_share_a_tree_with_different_settings():
    """the tree's guard used to end before the launch, so a second holder with another coarse_log2 could free
    the table a first holder's frame had already copied into its kernel parameters.  Two threads, two handles on one tree, one
    with the default table level and one a level coarser (each frame of one evicts the other's table and boxes): every frame of
    both must be the frame a lone caster renders."""
    import bench
    sc = bench.build_scene(9)
    w, h = 320, 200
    lone = _caster(sc, w, h)
    assert lone.compute()
    ref = lone.read_image().view(np.uint32).copy()
    lc = lone.memory_usage2()["coarse_log2"]
    a = _caster(sc, w, h)
    b = _caster(sc, w, h, tree_from=a, coarse_log2=lc - 1)
    assert a.memory_usage2()["tree_holders"] == 2
    bad, frames = [], 12

    def run(c, tag):
        for i in range(frames):
            if not c.compute():
                bad.append((tag, i, c.last_error()))
                return
            if not np.array_equal(c.read_image().view(np.uint32), ref):
                bad.append((tag, i, "frame differs"))

    ta, tb = threading.Thread(target=run, args=(a, "a")), threading.Thread(target=run, args=(b, "b"))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not bad, bad[:4]
