"""CPU restatement of the data path of EvImBuilder::Track (src/Event/EvImBuilder.cpp:1300-1515) out of the oracle's pieces.
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module (parity unpinned, like the
rest of oracle/: the reference holds no fixture for this path).

What is restated, per chunk of events, in the order the reference executes it:
  :1327-1343  calcEventGenRate (src/Event/EventData.cpp:14-19) / checkEvGenRate (:284-328)
  :1345       image = ev2im_gauss(l1Evs, W, H, sigma)
  :1348       makeFrame: state INIT -> EvBaseTracker::makeFrame = EvFrame with the detect-only ORBextractor ("FAST = ORB with one level",
              src/Event/EvBaseTracker.cpp:150-164); state TRACKING -> ELK_Tracker::trackAndMatchCurrImage (src/Event/KLT_Tracker.cpp:215-234:
              calcOpticalFlowPyrLK with the last tracked points as the initial flow, refineTrackedPts :104-151), median pixel displacement (:540-541)
  :1360-1379  INIT: init() (:568-592: ELK_Tracker::setRefImage) and the > DEF_TH_MIN_KPTS test
  :1380-1418  TRACKING: step() without the two-view refinement (geometry solver, out of scope: returns 1 as :609-611), updateState (:427-435),
              resolveEvWinSize (:209-232) with calcNewL1ChunkSize (:197-201)
  :1433-1470  dispatch: generateMCImage (:1146-1247) = the four reconstructions (getDPoseMCI :958-979, getBAMCI :1033-1058, getEvHist :1060-1079,
              getAff2DMCI :1124-1143), each measureImageFocus + cv::normalize; the winner by focus (std::multimap with std::greater<float>:
              the first inserted among equals, include/Utils/Visualization.h:29); "EH" rebuilt from the later half (:1214-1216);
              isMcImageGood (:260-267) = the L2 tracker's detection (2 x maxNumPts, src/Event/EvAsynchTracker.cpp:51) on the winner;
              the overlap handed back to the event queue (:1465-1469, resolveWinOverlap :234-239)
The poses of the motion-compensated reconstructions come from the optimisers (out of scope): the caller supplies them per window."""
import numpy as np

from . import oracle_py as orc

IDLE, INIT, TRACKING = 0, 1, 2
DEF_TH_MIN_KPTS, DEF_TH_MIN_MATCHES, DEF_L1_MAX_TRACK_LOST = 100, 50, 3      # include/Event/EventData.h:24-26, include/Event/EvImBuilder.h:39


def generate_mc_image(evs, W, H, sigma, cam, poses, l2_extractor, fast=False):
    """generateMCImage + isMcImageGood for the window `evs`; poses = dict(dp=, ba=, se2=) with absent methods None / missing.
    Returns dict(focus[5] (-1 = absent; [4] = the later-half histogram's), winner, image (u8), l2_kps)."""
    poses = poses or {}
    focus = np.full(5, -1.0, np.float32)
    imgs = {}
    for m, key in ((0, "dp"), (1, "ba")):
        p = poses.get(key)
        if p is not None:
            f32, _, _ = orc.ev2mci_se3(evs, cam, p["angle"], p["axis"], p["t"], p["medDepth"], W, H, sigma, False, False)
            focus[m] = orc.measure_image_focus(f32); imgs[m] = orc.cv_normalize_minmax_u8(f32)
    f32, _, _ = orc.ev2im_gauss(evs, W, H, sigma, False, False, fast=fast)
    focus[2] = orc.measure_image_focus(f32); imgs[2] = orc.cv_normalize_minmax_u8(f32)
    if poses.get("se2") is not None:
        f32, _, _ = orc.ev2mci_se2(evs, cam, poses["se2"], W, H, sigma, False, False)
        focus[3] = orc.measure_image_focus(f32); imgs[3] = orc.cv_normalize_minmax_u8(f32)
    # std::multimap<float, ..., std::greater<float>>::begin(): the largest key, the first inserted among equal keys (DP, BA, EH, Opt)
    winner = -1
    for m in (0, 1, 2, 3):
        if m in imgs and (winner < 0 or focus[m] > focus[winner]):
            winner = m
    pref = len(evs) // 2                                  # getEvHist(evs, pMCI, imSTDEH, evs.size() / 2) :1214-1216, window :1062-1067
    half = evs[len(evs) - pref:] if pref > 0 else evs
    hf32, _, _ = orc.ev2im_gauss(half, W, H, sigma, False, False, fast=fast)
    focus[4] = orc.measure_image_focus(hf32)
    image = orc.cv_normalize_minmax_u8(hf32) if winner == 2 else imgs[winner]
    _, l2_kps, _, _ = l2_extractor.extract(image, (0, 1000), False)
    return dict(focus=focus, winner=winner, image=image, l2_kps=l2_kps)


class L1Chain:
    def __init__(self, W=240, H=180, l1ChunkSize=2000, l1NumLoop=3, l1FixedWinSz=False, continTracking=True, maxPixelDisp=3.0,
                 l1WinOverlap=0.5, minEvGenRate=1.0, l1ImSigma=1.0, maxNumPts=400, fastTh=0, imMargin=9, klt=(23, 1, 10, 0.03), cam=None,
                 fast=False):
        self.W, self.H, self.sigma, self.cam, self.fast = W, H, float(l1ImSigma), cam, fast
        self.cont, self.fixed = bool(continTracking), bool(l1FixedWinSz)
        self.win = self.win0 = int(l1ChunkSize)
        self.num_loop, self.max_disp, self.overlap, self.min_rate = int(l1NumLoop), float(maxPixelDisp), float(l1WinOverlap), float(minEvGenRate)
        self.n_overlap_fixed = int(self.overlap * float(self.num_loop * self.win))
        self.klt = klt
        self.l1 = orc.OrbExtractor(maxNumPts, 1.0, 1, fastTh, 0, edgeTh=imMargin, imWidth=W, fast=fast)
        self.l2 = orc.OrbExtractor(2 * maxNumPts, 1.0, 1, fastTh, 0, edgeTh=imMargin, imWidth=W, fast=fast)
        self.state = IDLE
        self._reset()

    def _reset(self):
        self.idx = 0; self.low = 0; self.acc = []; self.cnt = None

    def _check_rate(self, rate):
        if rate > self.min_rate:
            self.low = 0
            return 0
        if self.cont:
            return -1 if (not self.fixed and self.state == INIT) else 0
        if self.state == INIT:
            return -1
        self.low += 1
        return -1 if self.low > DEF_L1_MAX_TRACK_LOST else 1

    def track(self, evs, mci_poses=None):
        out = dict(state=None, dispatched=False)
        if self.state == IDLE:
            self.state = INIT
        if self.state == INIT:
            self._reset()
        if len(evs) == 0:
            return out
        span = float(evs["ts"][-1]) - float(evs["ts"][0])
        with np.errstate(divide="ignore", invalid="ignore"):
            rate = np.float64(len(evs)) / (np.float64(span) * self.W * self.H)
        out["state"] = self.state
        r = self._check_rate(rate)
        if r != 0:
            if r == -1:
                self.state = INIT
            else:
                self.acc.append(evs)
            out["skipped"] = r
            return out
        _, image, _ = orc.ev2im_gauss(evs, self.W, self.H, self.sigma, False, True, fast=self.fast)
        out["image"] = image
        send = False
        if self.state == INIT:
            _, kps, _, _ = self.l1.extract(image, (0, 1000), False)
            out["kps"] = kps
            self.cnt = np.ones(len(kps), np.int32)
            self.ref_img, self.ref_kps = image, kps
            self.ref_pts = np.stack([kps["x"], kps["y"]], axis=1).astype(np.float32)
            self.last_pts = self.ref_pts.copy()
            if len(kps) > DEF_TH_MIN_KPTS or (self.cont and self.fixed):
                self.idx += 1; self.acc.append(evs)
                self.state = TRACKING
            else:
                self.win = self.win0
        else:
            win, lev, itr, eps = self.klt
            pts, st, err = orc.calc_optical_flow_pyr_lk(self.ref_img, image, self.ref_pts, self.last_pts, win, lev, itr, eps, flags=4)
            self.last_pts = pts
            n = len(self.ref_pts)
            ok = np.zeros(n, bool)
            disp = []
            for i in range(n):                                    # refineTrackedPts :134-151
                x, y = pts[i]
                if st[i] == 1 and x >= 0 and x < np.float32(self.W) and y >= 0 and y < np.float32(self.H):
                    ok[i] = True
                    dx = np.float32(x - self.ref_pts[i, 0]); dy = np.float32(y - self.ref_pts[i, 1])
                    disp.append(np.sqrt(np.float32(np.float32(dx * dx) + np.float32(dy * dy))))
            self.cnt[ok] += 1
            m12 = np.where(ok, np.arange(n), -1).astype(np.int32)
            disp.sort()
            med = float(disp[len(disp) // 2]) if disp else 0.0
            nm = int(ok.sum())
            out.update(pts=pts, status=st, err=err, matches12=m12, nMatches=nm, medPxDisp=med)
            if nm < DEF_TH_MIN_MATCHES and not self.cont and self.idx < 3:
                self.win = self.win0; self.state = INIT
                return out
            self.idx += 1; self.acc.append(evs)
            dispatch = False
            if not self.fixed and np.float32(med) > np.float32(self.max_disp):
                self.win = int(np.floor(np.float32(np.float32(self.idx + 1) / np.float32(med)) * np.float32(self.win)))
                dispatch = True
            elif self.fixed and self.idx >= self.num_loop:
                dispatch = True
            if dispatch or nm < DEF_TH_MIN_MATCHES:
                self.state = INIT
                send = True
        out["chunk_size"] = self.win
        if send:
            acc = np.concatenate(self.acc)
            mc = generate_mc_image(acc, self.W, self.H, self.sigma, self.cam, mci_poses(acc) if mci_poses else None, self.l2, self.fast)
            out.update(dispatched=True, mci=mc, mci_good=(len(mc["l2_kps"]) > DEF_TH_MIN_KPTS or self.cont), window=len(acc))
            if self.cont:
                nov = self.n_overlap_fixed if self.fixed else int(len(acc) * self.overlap)
                out["overlap"] = acc[len(acc) - nov:]
        return out


def run_sequence(builder_track, events, chunk0, mci_poses=None, max_chunks=10 ** 9):
    """The track manager's feed (src/Event/EvTrackManager.cpp:272-286 consumeEventsBegin + :1465-1469 injectEventsBegin): chunks of the
    builder's CURRENT l1ChunkSize off the front of the queue, an overlap handed back goes to the front again.  builder_track(chunk,
    mci_poses) -> the chunk's result dict with `chunk_size` = the size to use next.  Returns the list of results."""
    queue = events
    size = int(chunk0)
    res = []
    while len(queue) >= max(size, 1) and len(res) < max_chunks:
        chunk, queue = queue[:size], queue[size:]
        r = builder_track(chunk, mci_poses)
        res.append(r)
        if r.get("overlap") is not None and len(r["overlap"]):
            queue = np.concatenate([r["overlap"], queue])
        size = int(r.get("chunk_size", size))
        if size < 1:
            break
    return res
