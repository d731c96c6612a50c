"""Reader and comparator of the per-frame output log (XRSLAM_AMD_DUMP_OUT, xrslam_amd/csrc/host/ba_dump.hpp: OutLogger).

north_star's output list -- "track IDs, keypoint pixel positions, pose/velocity/bias states ... bit-exact for indices and within
1e-4 relative on floats" -- per frame, for two builds of the pipeline run on the same stream: `compare()` requires every id,
index, count and tag to be EQUAL and every float to be within the tolerance, and reports how many of the floats were bit-identical."""
import struct

import numpy as np

TT_VALID, TT_TRIANGULATED, TT_FIX_INVD, TT_TRASH, TT_STATIC, TT_OUTLIER = range(6)


def read(path):
    """-> (F, B): lists of dicts in file order (one 'F' per frame tracked, one 'B' per frame the sliding-window tracker processed)."""
    F, B = [], []
    with open(path, "rb") as fh:
        data = fh.read()
    o = 0
    while o + 5 <= len(data):
        tag, n = struct.unpack_from("<BI", data, o)
        o += 5
        p = data[o:o + n]
        assert len(p) == n, "truncated record"
        o += n
        if tag == ord("F"):
            fid, t, k = struct.unpack_from("<QdI", p, 0)
            rec = np.frombuffer(p, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("track", "<i8")]), count=k, offset=20)
            F.append({"id": fid, "t": t, "px": np.stack([rec["x"], rec["y"]], 1), "track": rec["track"].copy()})
        elif tag == ord("B"):
            q = 0
            fid, kf = struct.unpack_from("<QI", p, q)
            q += 12
            state = np.frombuffer(p, "<f8", 16, q).copy()
            q += 128
            (nw,) = struct.unpack_from("<I", p, q)
            q += 4
            window = []
            for _ in range(nw):
                wid, ns = struct.unpack_from("<QI", p, q)
                q += 12
                subs = list(struct.unpack_from("<%dQ" % ns, p, q)) if ns else []
                q += 8 * ns
                window.append((wid, subs))
            (nk,) = struct.unpack_from("<I", p, q)
            q += 4
            kp_track = np.frombuffer(p, "<i8", nk, q).copy()
            q += 8 * nk
            (nt,) = struct.unpack_from("<I", p, q)
            q += 4
            tr = np.frombuffer(p, dtype=np.dtype([("id", "<u8"), ("tags", "<u4"), ("inv_depth", "<f8"), ("x", "<f8"), ("y", "<f8"),
                                                   ("z", "<f8")]), count=nt, offset=q)
            B.append({"id": fid, "keyframe": kf, "state": state, "window": window, "kp_track": kp_track,
                      "track_id": tr["id"].copy(), "track_tags": tr["tags"].copy(), "inv_depth": tr["inv_depth"].copy(),
                      "point": np.stack([tr["x"], tr["y"], tr["z"]], 1)})
        else:
            raise ValueError("unknown record tag %r" % tag)
    return F, B


def compare(got, want, rtol=1e-4, atol_px=1e-3, atol_state=1e-6, atol_point=1e-4):
    """got / want: read() of two runs.  Asserts; returns statistics (how many floats were bit-identical)."""
    (Fg, Bg), (Fw, Bw) = got, want
    assert len(Fg) == len(Fw) and len(Bg) == len(Bw), (len(Fg), len(Fw), len(Bg), len(Bw))
    st = {"frames": len(Fw), "backend_frames": len(Bw), "keypoints": 0, "keypoints_bit_identical": 0, "tracks_compared": 0,
          "landmarks": 0, "landmarks_bit_identical": 0, "states_bit_identical": 0, "max_px_diff": 0.0, "max_state_rel": 0.0}
    for a, b in zip(Fg, Fw):
        assert a["id"] == b["id"] and a["t"] == b["t"]
        assert a["px"].shape == b["px"].shape, "frame %d: %d vs %d key points" % (b["id"], len(a["px"]), len(b["px"]))
        np.testing.assert_array_equal(a["track"], b["track"], err_msg="track ids of frame %d" % b["id"])
        np.testing.assert_allclose(a["px"], b["px"], rtol=rtol, atol=atol_px, err_msg="key-point pixels of frame %d" % b["id"])
        st["keypoints"] += len(b["px"])
        st["keypoints_bit_identical"] += int((a["px"] == b["px"]).all(1).sum())
        if len(b["px"]):
            st["max_px_diff"] = max(st["max_px_diff"], float(np.abs(a["px"] - b["px"]).max()))
    for a, b in zip(Bg, Bw):
        assert a["id"] == b["id"] and a["keyframe"] == b["keyframe"], "backend record of frame %d" % b["id"]
        assert a["window"] == b["window"], "window after frame %d" % b["id"]
        np.testing.assert_array_equal(a["kp_track"], b["kp_track"], err_msg="window-map tracks of frame %d" % b["id"])
        np.testing.assert_array_equal(a["track_id"], b["track_id"], err_msg="window-map track list after frame %d" % b["id"])
        np.testing.assert_array_equal(a["track_tags"], b["track_tags"], err_msg="track tags after frame %d" % b["id"])
        # pose / velocity / biases: 1e-4 relative on each block's magnitude (a bias component of 1e-5 is not held to 1e-9)
        for lo, hi in ((0, 4), (4, 7), (7, 10), (10, 13), (13, 16)):
            scale = max(np.linalg.norm(b["state"][lo:hi]), 1e-2)
            err = np.abs(a["state"][lo:hi] - b["state"][lo:hi]).max()
            assert err <= rtol * scale + atol_state, "state[%d:%d] of frame %d: %g" % (lo, hi, b["id"], err)
            st["max_state_rel"] = max(st["max_state_rel"], float(err / scale))
        st["states_bit_identical"] += int((a["state"] == b["state"]).all())
        tri = (b["track_tags"] >> TT_TRIANGULATED) & 1 == 1
        np.testing.assert_allclose(a["inv_depth"][tri], b["inv_depth"][tri], rtol=rtol, atol=1e-7,
                                   err_msg="inverse depths after frame %d" % b["id"])
        ok = tri & ((b["track_tags"] >> TT_VALID) & 1 == 1)
        np.testing.assert_allclose(a["point"][ok], b["point"][ok], rtol=rtol, atol=atol_point, err_msg="landmarks after frame %d" % b["id"])
        st["tracks_compared"] += len(b["track_id"])
        st["landmarks"] += int(ok.sum())
        st["landmarks_bit_identical"] += int((a["point"][ok] == b["point"][ok]).all(1).sum())
    return st
