"""An independent model of the sliding-window tracker's structural decisions, small enough to audit against the
reference by eye (core/sliding_window_tracker.cpp): which frames become keyframes, which hang below a keyframe as
subframes, how a subframe is lifted or re-attached when the motion changes between "translating" and "rotation only",
how subframe triples are merged while the camera only rotates, and which keyframe leaves the window.

It is NOT the product's code path (that is xrslam_amd/csrc/host/pipeline.hpp, C++): it is fed with the inputs the C++
tracker logged at every frame (XRSLAM_AMD_DUMP_SWT, ba_dump.hpp) and must arrive at the same window -- frame ids, tags and
subframe lists -- after every frame.  Test infrastructure.

A window entry is [frame_id, no_translation, [[sub_id, sub_no_translation], ...]]."""


class SwtModel:
    def __init__(self, window, size, subframe_size, force_keyframe_landmarks):
        self.window = [[f, nt, [list(s) for s in subs]] for f, nt, subs in window]
        self.size, self.subframe_size, self.force = size, subframe_size, force_keyframe_landmarks

    def step(self, frame_id, no_translation, mapped):
        """One SlidingWindowTracker::track() after mirror_frame attached the new frame (:82-117).  Returns is_keyframe."""
        w = self.window
        w.append([frame_id, no_translation, []])
        is_kf = self._manage_keyframe(mapped)
        if is_kf:
            while len(w) > self.size:                     # slide_window (:360-368): the oldest keyframe goes, with its subframes
                w.pop(0)
        else:
            self._merge_rotation_subframes()              # the structural part of refine_subwindow (:370-393)
        return is_kf

    def _manage_keyframe(self, mapped):                   # :145-223
        w = self.window
        kf_i, nf_j = w[-2], w[-1]
        subs = kf_i[2]
        if subs:
            if subs[-1][1]:                               # the last subframe saw rotation only
                if not nf_j[1]:                           # ... and the new frame translates: that subframe becomes a keyframe
                    lifted = subs.pop()                   # in front of the new frame, which becomes a keyframe too (:156-168)
                    w.insert(len(w) - 1, [lifted[0], lifted[1], []])
                    return True
                # both rotation only: fall through to the landmark count
            else:
                if nf_j[1]:                               # translating subframes, rotating new frame: the last subframe is lifted
                    lifted = subs.pop()                   # to a keyframe and the new frame hangs below it (:171-184)
                    w.pop()
                    w.append([lifted[0], lifted[1], [[nf_j[0], nf_j[1]]]])
                    return True
                if len(subs) >= self.subframe_size:       # enough subframes: the new frame is a keyframe (:186-195)
                    return True
        if mapped < self.force:                           # too few mapped landmarks in view (:199-210)
            return True
        w.pop()                                           # otherwise the new frame hangs below the last keyframe (:218-221)
        kf_i[2].append([nf_j[0], nf_j[1]])
        return False

    def _merge_rotation_subframes(self):                  # :375-392
        frame = self.window[-1]
        subs = frame[2]
        if not subs or not subs[0][1]:
            return
        if len(subs) >= 9:
            for i in range(len(subs) // 3, 0, -1):
                for j in range(i * 3 - 1, (i - 1) * 3, -1):
                    del subs[j - 1]


# ------------------------------------------------------------------------------------------------ landmark decisions
# track_landmark (core/sliding_window_tracker.cpp:225-245 -> map/track.cpp:46-76, 97-101, geometry/stereo.h:84-94) and the
# landmark sweep after the window solve (:325-357), restated with numpy from the reference; fed with the observations the C++
# tracker logged (XRSLAM_AMD_DUMP_SWT: camera pose q xyzw, p; bearing; keyframe flag; fx fy cx cy -- in the track's order).
import numpy as np  # noqa: E402


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def triangulate(obs):
    """Track::triangulate: DLT over every observation, then the cheirality test q_z * h_w > 0 in every view.
    Returns the point or None."""
    A, Ps = [], []
    for o in obs:
        R = _rot(o[:4]).T                                 # camera-from-world
        P = np.hstack([R, (-R @ np.array(o[4:7])).reshape(3, 1)])
        z = o[7:10]
        A.append(z[0] * P[2] - z[2] * P[0])
        A.append(z[1] * P[2] - z[2] * P[1])
        Ps.append(P)
    h = np.linalg.svd(np.array(A))[2][-1]                 # right singular vector of the smallest singular value
    for P in Ps:
        if not (P[2] @ h) * h[3] > 0:
            return None
    return h[:3] / h[3]


def anchor_inv_depth(obs, p):
    """Track::set_landmark_point: the inverse distance from the track's first observation."""
    o = obs[0]
    return 1.0 / np.linalg.norm(_rot(o[:4]).T @ (p - np.array(o[4:7])))


def landmark_point(obs, inv_depth):
    """Track::get_landmark_point."""
    o = obs[0]
    return _rot(o[:4]) @ np.array(o[7:10]) / inv_depth + np.array(o[4:7])


def landmark_is_valid(obs, inv_depth):
    """The sweep of refine_window over a triangulated track: depth in (1e-3, 50] in every KEYFRAME that sees it, and a mean
    reprojection error below 3 pixels over those keyframes."""
    x = landmark_point(obs, inv_depth)
    rpe, cnt = 0.0, 0.0
    for o in obs:
        if not o[10]:
            continue
        y = _rot(o[:4]).T @ (x - np.array(o[4:7]))
        if y[2] <= 1.0e-3 or y[2] > 50:
            return False
        fx, fy, cx, cy = o[11:15]
        z = o[7:10]
        a = np.array([y[0] / y[2] * fx + cx, y[1] / y[2] * fy + cy])
        b = np.array([z[0] / z[2] * fx + cx, z[1] / z[2] * fy + cy])
        rpe += np.linalg.norm(a - b)
        cnt += 1.0
    return rpe / max(cnt, 1.0) < 3.0
