"""Bounding boxes of a page -> region rows in the vector store (the step feeding the hot path).

Mirrors `RegionProcessor.process_image_regions` (deprecated_package/region_processor.py:62-158)
and the cache schema of `DocLayoutDetector.detect_regions` (doclayout_detector.py:145-153).  The
reference re-opens and re-decodes the full page PNG once PER REGION
(`get_region_image`, doclayout_detector.py:178), writes every crop to a PNG and re-reads it in the
embedder; here the page is decoded once, uploaded once, all boxes are cut on the device (K0
`mme_crop_boxes`) and go straight into the preprocessing + encoder kernels.
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np

from . import config
from ._lib import MmeError

logger = logging.getLogger(__name__)


def load_region_cache(path):
    """One `region_cache/*.json` file (doclayout_detector.py:145-153): dict with `boxes`, `classes`,
    `scores`, `class_names`, `image_size {width, height}`."""
    with open(path) as fh:
        regions = json.load(fh)
    for key in ("boxes", "classes", "class_names", "scores"):
        regions.setdefault(key, [])
    regions.setdefault("image_size", {"width": 0, "height": 0})
    return regions


def region_rows(image_path, regions, region_types=None):
    """The rows region_processor.py:75-113 builds for one page, without touching pixels.

    Returns (ids, metadatas, int_boxes int32[n,4]) for the boxes whose class is embedded
    (:76-77), in detector order; `int_boxes` are the `map(int, box)` corners (:88) that define both
    the crop (doclayout_detector.py:179) and `area_percentage` (:89-93)."""
    region_types = config.REGION_TYPES_TO_PROCESS if region_types is None else region_types
    image_filename = os.path.basename(image_path)
    image_size = regions.get("image_size", {"width": 0, "height": 0})
    ids, metas, boxes = [], [], []
    for i, (box, class_id, class_name, score) in enumerate(
        zip(regions.get("boxes", []), regions.get("classes", []), regions.get("class_names", []), regions.get("scores", []))
    ):
        if class_name not in region_types:
            continue
        x_min, y_min, x_max, y_max = map(int, box)
        region_width, region_height = x_max - x_min, y_max - y_min
        total_area = image_size["width"] * image_size["height"]
        area_percentage = (region_width * region_height / total_area) * 100 if total_area else 0
        ids.append(f"region_{os.path.splitext(image_filename)[0]}_{i}")
        metas.append({
            "parent_image": image_path,
            "parent_image_name": image_filename,
            "region_index": i,
            "region_type": class_name,
            "region_class_id": int(class_id),
            "region_score": float(score),
            "box": ",".join(map(str, box)),
            "box_normalized": ",".join(map(str, [x_min / image_size["width"], y_min / image_size["height"],
                                                 x_max / image_size["width"], y_max / image_size["height"]])) if total_area else "",
            "area_percentage": area_percentage,
            "width": region_width,
            "height": region_height,
            "is_region": True,
        })
        boxes.append([x_min, y_min, x_max, y_max])
    return ids, metas, np.asarray(boxes, dtype=np.int32).reshape(-1, 4)


class CachedRegionDetector:
    """The cache half of `DocLayoutDetector.detect_regions` (doclayout_detector.py:99-120): the regions of a page come from
    `<cache_folder>/<page-stem>_conf<c>_iou<i>.json` when that file exists.  Detection itself (DocLayout-YOLO) is out of
    scope: a page without a cache file has no regions here (None, as the reference returns on failure)."""

    def __init__(self, cache_folder, conf_threshold=0.1, iou_threshold=0.45):
        self.cache_folder, self.conf_threshold, self.iou_threshold = cache_folder, conf_threshold, iou_threshold

    def cache_path(self, image_path):
        stem = os.path.splitext(os.path.basename(image_path))[0]
        return os.path.join(self.cache_folder, f"{stem}_conf{self.conf_threshold}_iou{self.iou_threshold}.json")

    def detect_regions(self, image_path, force_recompute=False):
        path = self.cache_path(image_path)
        if not os.path.exists(path):
            logger.error(f"Error detecting regions in {os.path.basename(image_path)}: no cached regions at {path}")
            return None
        try:
            regions = load_region_cache(path)
            logger.info(f"Loaded cached regions for {os.path.basename(image_path)}")
            return regions
        except Exception as e:  # noqa: BLE001
            logger.warning(f"Error loading cached regions: {e}.")
            return None


class RegionProcessor:
    """Same constructor shape and entry points as region_processor.py:RegionProcessor.

    `embedder` is a `RegionEmbedder`; `collection` anything with chroma's `upsert`.  `detector` is only asked for
    `detect_regions(image_path, force_recompute)` by `process_regions` (a `CachedRegionDetector`, or the reference's
    own detector object): crops are never cut by it, they are cut on the GPU from the decoded page."""

    WAVE_CROPS = 1024       # crops a device pass of `process_regions` should carry at least (the encoder's rate per crop
    WAVE_BYTES = 1 << 30    # flattens out from ~1k crops per pass) ... and at most this many packed pixel bytes
    DECODE_AHEAD = 4        # pages decoded ahead of the device, on threads (Pillow releases the GIL while decoding)
    WAVE_SLOTS = 3          # page arenas: one being cut on the device, one staged and waiting, one being uploaded
    WAVE_PAGE_BYTES = 2 << 30  # page pixels a wave may hold on the device (pages of few boxes close a wave by this)

    def __init__(self, embedder, collection, detector=None):
        self.embedder, self.collection, self.detector = embedder, collection, detector

    def _page_to_device(self, page):
        from .embedder import _load_rgb

        t = self.embedder.torch
        arr = page if isinstance(page, np.ndarray) and page.dtype == np.uint8 and page.ndim == 3 else _load_rgb(page)
        dev = t.device(f"cuda:{self.embedder.engine.device}")
        return t.from_numpy(np.require(arr, requirements=["C", "W"])).to(dev)

    def embed_page_regions(self, page, int_boxes):
        """page: path | PIL image | uint8[H,W,3]; int_boxes int32[n,4] -> float32 CUDA tensor [n, 768].

        Boxes of zero or negative size (which make the reference's PNG save fail, :115-117) raise."""
        pix, offs, hw = self.embedder.engine.crop_boxes(self._page_to_device(page), int_boxes)
        e32, _ = self.embedder.embed_packed(pix, offs, hw, want_bf16=False)
        return e32

    def _page_rows(self, image_path, regions):
        """Rows of one page that will be embedded: (ids, metas, boxes, good) with the reference's warnings (:85-87)."""
        ids, metas, boxes = region_rows(image_path, regions)
        good = [k for k in range(len(ids)) if boxes[k, 2] > boxes[k, 0] and boxes[k, 3] > boxes[k, 1]]
        for k in sorted(set(range(len(ids))) - set(good)):
            logger.warning(f"Failed to extract region {metas[k]['region_index']} from {os.path.basename(image_path)}")
        return ids, metas, boxes, good

    def _upsert_page(self, image_filename, ids, metas, good, emb):
        """:124-154: the page's rows go to the store in chunks of REGION_BATCH_SIZE, in detector order."""
        embedded_count = 0
        for i in range(0, len(good), config.REGION_BATCH_SIZE):
            sel = good[i : i + config.REGION_BATCH_SIZE]
            batch_meta = [metas[k] for k in sel]
            documents = [f"Region: {m['region_type']} from {m['parent_image_name']}" for m in batch_meta]
            try:
                self.collection.upsert(ids=[ids[k] for k in sel], embeddings=emb[i : i + len(sel)], documents=documents,
                                       metadatas=batch_meta)
                embedded_count += len(sel)
                logger.info(f"Embedded {len(sel)} regions from {image_filename}")
            except Exception as e:  # :153-154
                logger.error(f"DB Error: {e}")
        return embedded_count

    def process_image_regions(self, image_path, regions, page=None):
        """region_processor.py:62-158: embed the page's regions and upsert them; returns the count.

        `page` overrides the pixels read from `image_path` (already decoded page).  One device pass per page: a caller
        with many pages should use `process_regions`, which fills the passes across pages."""
        image_filename = os.path.basename(image_path)
        if not regions.get("boxes"):
            return 0
        ids, metas, boxes, good = self._page_rows(image_path, regions)
        if not good:
            return 0
        try:
            emb = self.embed_page_regions(image_path if page is None else page, boxes[good]).cpu().tolist()
        except (MmeError, OSError, ValueError) as e:
            logger.error(f"Error in batch processing: {e}")  # embedder.py:223-224: every region of the page fails
            return 0
        return self._upsert_page(image_filename, ids, metas, good, emb)

    def _wave_pipe(self):
        """Staging of `process_regions`, created once per processor: a copy stream for the page uploads, a result stream,
        WAVE_SLOTS device arenas holding the pages of a wave, one packed-crop buffer, two pinned result buffers and the
        events that order them."""
        if getattr(self, "_pipe", None) is None:
            from .embedder import _Null

            t = self.embedder.torch
            dev = t.device(self.embedder.device)
            cuda = dev.type == "cuda"
            mk_stream, mk_event = ((lambda: t.cuda.Stream(dev)), t.cuda.Event) if cuda else (_Null, _Null)
            ns = max(2, int(self.WAVE_SLOTS))
            self._pipe = {"dev": dev, "cuda": cuda, "in_stream": mk_stream(), "out_stream": mk_stream(),
                          "arena": [None] * ns, "ev_cut": [mk_event() for _ in range(ns)], "slot_used": [False] * ns,
                          "pix": None, "out": [None, None], "ev_pass": [mk_event(), mk_event()], "ev_out": [mk_event(), mk_event()]}
        return self._pipe

    def process_regions(self, image_paths, force_recompute=False, *, regions_by_path=None, pages=None, as_lists=True):
        """region_processor.py:36-60 `process_regions(image_paths, force_recompute)`: every page's regions embedded and
        upserted, page by page in `image_paths` order; returns the number of regions processed.

        The reference calls `process_image_regions` per page, i.e. the embedder sees 5 ... 220 crops at a time (<= 48 per
        call, :124-129) -- a device pass of this engine reaches its rate from ~1k crops on.  Here the boxes of SEVERAL
        pages fill one pass ("wave"), and the host side runs under the device passes:
          * pages are decoded ahead on threads (Pillow releases the GIL);
          * a producer thread uploads the pages of a wave into one of WAVE_SLOTS device arenas on a copy stream (DMA: it
            runs beside the encoder's kernels); a wave closes at >= WAVE_CROPS crops (or WAVE_BYTES of crop pixels, or
            WAVE_PAGE_BYTES of page pixels);
          * this thread cuts the boxes of the wave's pages on the device (K0, straight into the packed-crop buffer) and
            runs ONE `mme_embed` over them -- both on the compute stream: a K0 launch of its own stream would wait for
            a gap between the encoder's persistent kernels, page after page -- copies the rows back on a third stream
            and hands the rows of the previous wave to the store -- per page, in page order, in the reference's chunks
            of REGION_BATCH_SIZE -- while the device works on the current one.
        The contract per page is unchanged: an unreadable page is logged and skipped (`validate_image`, :43-45), a page
        without regions is skipped with the reference's warning (:50-52), a page whose boxes cannot be cut fails alone and
        a failed device pass voids exactly the pages it carried ("Error in batch processing"), a failed upsert voids its
        chunk.

        `regions_by_path` (path -> regions dict) replaces `detector.detect_regions`; `pages` (path -> decoded uint8
        [H, W, 3] array) replaces reading the file; `as_lists=False` upserts float32 ndarray rows instead of float lists.
        The passes run on the embedder's FIRST context (one GPU): the deployment shape is one process per GPU, each with its
        share of the pages (`dist.shard_range` / `shard_pages`); an embedder that holds several contexts in one process fans
        `get_image_embeddings` out over them, not this method."""
        import contextlib
        import queue
        import threading
        import time as _time
        from concurrent.futures import ThreadPoolExecutor

        from .embedder import _load_rgb

        t = self.embedder.torch
        engine = self.embedder.engine
        pipe = self._wave_pipe()
        dev, cuda = pipe["dev"], pipe["cuda"]
        ns = len(pipe["arena"])
        total_images = len(image_paths)
        logger.info(f"Processing regions for {total_images} images")
        on = (lambda stream: t.cuda.stream(stream)) if cuda else (lambda stream: contextlib.nullcontext())
        trace = getattr(self, "trace", None)  # tools: a dict of accumulated seconds per stage of the two threads

        def tick(key, t0):
            if trace is not None:
                trace[key] = trace.get(key, 0.0) + (_time.perf_counter() - t0)
            return _time.perf_counter()

        def decode(path):
            try:
                if pages is not None and path in pages:
                    return pages[path]
                return _load_rgb(path)
            except Exception as e:  # image_utils.py:26-35 validate_image
                logger.error(f"Invalid image file {path}: {e}")
                return None

        ready = queue.Queue(maxsize=ns)
        slot_free = [threading.Semaphore(1) for _ in range(ns)]
        stop = threading.Event()

        def produce():
            wave_no = 0
            cur = None  # open wave: {"slot", "pages": [(path, ids, metas, good, boxes, arena offset, shape)], "crops", "crop_bytes", "page_bytes"}

            def open_wave():
                nonlocal cur
                slot = wave_no % ns
                while not slot_free[slot].acquire(timeout=0.1):
                    if stop.is_set():
                        return False
                if pipe["slot_used"][slot]:
                    with on(pipe["in_stream"]):
                        pipe["in_stream"].wait_event(pipe["ev_cut"][slot])  # the K0 launches that read this arena have run
                pipe["slot_used"][slot] = True
                cur = {"slot": slot, "pages": [], "crops": 0, "crop_bytes": 0, "page_bytes": 0}
                return True

            def close_wave():
                nonlocal cur, wave_no
                if cur is None:
                    return
                if cur["pages"]:
                    ready.put(("wave", cur))  # every upload of the wave has completed: the copies below are blocking
                    wave_no += 1
                else:
                    slot_free[cur["slot"]].release()
                cur = None

            def upload(arr, slot, at):
                """Page pixels -> the wave's arena at byte `at`, straight from the decoded (pageable) array: on this platform
                the driver's own staging moves pageable memory at the pinned rate, idle or beside the encoder
                (tools/host_copy_probe.py: 56 GB/s), so a pinned copy in front of the DMA would only cost a pass over host
                memory.  Returns when the copy has been made."""
                arr = np.require(arr, requirements=["C"])
                nb = arr.nbytes
                arena = pipe["arena"][slot]
                if arena is None or arena.numel() < at + nb:
                    grown = t.empty(max(at + nb, self.WAVE_PAGE_BYTES // 2, 1 << 24), dtype=t.uint8, device=dev)
                    if arena is not None:
                        if at:  # pages of this wave already sit in the old arena
                            with on(pipe["in_stream"]):
                                grown[:at].copy_(arena[:at])
                                pipe["in_stream"].synchronize()
                        pipe["ev_cut"][slot].synchronize()  # nothing reads the old arena any more
                    pipe["arena"][slot] = arena = grown
                with on(pipe["in_stream"]):
                    arena[at : at + nb].copy_(t.from_numpy(arr.reshape(-1)))
                return nb

            try:
                with ThreadPoolExecutor(max_workers=self.DECODE_AHEAD) as pool:
                    futures = {k: pool.submit(decode, image_paths[k]) for k in range(min(self.DECODE_AHEAD, total_images))}
                    for idx, image_path in enumerate(image_paths):
                        if stop.is_set():
                            return
                        tk = _time.perf_counter()
                        arr = futures.pop(idx).result()
                        tk = tick("producer: wait for decode", tk)
                        if idx + self.DECODE_AHEAD < total_images:
                            futures[idx + self.DECODE_AHEAD] = pool.submit(decode, image_paths[idx + self.DECODE_AHEAD])
                        if (idx + 1) % 5 == 0 or idx == total_images - 1:
                            logger.info(f"Region processing progress: {idx + 1}/{total_images} images")
                        if arr is None:
                            logger.error(f"Skipping invalid image: {image_path}")  # :43-45
                            continue
                        image_filename = os.path.basename(image_path)
                        try:
                            regions = regions_by_path.get(image_path) if regions_by_path is not None else self.detector.detect_regions(image_path, force_recompute)
                        except Exception as e:  # noqa: BLE001
                            logger.error(f"Error detecting regions in {image_filename}: {e}")
                            regions = None
                        if regions is None or not regions.get("boxes"):
                            logger.warning(f"No regions detected in {image_filename}")  # :50-52
                            continue
                        ids, metas, boxes, good = self._page_rows(image_path, regions)
                        tk = tick("producer: rows", tk)
                        if not good:
                            continue
                        b = boxes[good]
                        side = np.stack([b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]], axis=1)
                        if side.max() > 8000:  # mme_crop_boxes' limit (the embedder's own 8000-px cap, embedder.py:110-114)
                            logger.error(f"Error in batch processing: a region of {image_filename} is {int(side.max())} px long (supported: 8000)")
                            continue
                        nbytes = int(((side[:, 0].astype(np.int64) * side[:, 1] * 3 + 15) // 16 * 16).sum())
                        if cur is not None and cur["pages"] and (cur["crop_bytes"] + nbytes > self.WAVE_BYTES
                                                                 or cur["page_bytes"] + arr.nbytes > self.WAVE_PAGE_BYTES):
                            close_wave()
                        if cur is None and not open_wave():
                            return
                        tk = tick("producer: wait for a wave slot", tk)
                        try:
                            at = (cur["page_bytes"] + 255) // 256 * 256
                            nb = upload(arr, cur["slot"], at)
                        except (OSError, ValueError, RuntimeError) as e:
                            logger.error(f"Error in batch processing: {e}")  # this page's regions fail, the wave goes on
                            continue
                        tk = tick("producer: page H2D", tk)
                        cur["pages"].append((image_path, ids, metas, good, b, at, arr.shape))
                        cur["page_bytes"] = at + nb
                        cur["crop_bytes"] += nbytes
                        cur["crops"] += len(good)
                        if cur["crops"] >= self.WAVE_CROPS:
                            close_wave()
                    close_wave()
            except BaseException as e:  # never leave the consumer waiting
                logger.error(f"Error in batch processing: {e}")
                if cur is not None:
                    slot_free[cur["slot"]].release()
            finally:
                ready.put(("end",))

        total = 0
        pending = None  # (result slot, wave, n rows, device rows kept alive) whose D2H is in flight
        n_pass = 0

        def finalize(p):
            nonlocal total
            rs, wave, n, _keep = p
            pipe["ev_out"][rs].synchronize()
            rows = pipe["out"][rs][:n].numpy()
            r0 = 0
            for path, ids, metas, good, *_ in wave["pages"]:
                emb = rows[r0: r0 + len(good)]
                r0 += len(good)
                total += self._upsert_page(os.path.basename(path), ids, metas, good, emb.tolist() if as_lists else list(emb.copy()))

        producer = threading.Thread(target=produce, name="mme-region-stage", daemon=True)
        producer.start()
        try:
            compute = t.cuda.current_stream(dev) if cuda else pipe["in_stream"]
            while True:
                tk = _time.perf_counter()
                msg = ready.get()
                tk = tick("consumer: wait for a staged wave", tk)
                if msg[0] == "end":
                    break
                wave = msg[1]
                slot = wave["slot"]
                try:
                    need = wave["crop_bytes"] + 16
                    if pipe["pix"] is None or pipe["pix"].numel() < need:
                        pipe["pix"] = t.empty(max(need, self.WAVE_BYTES // 4), dtype=t.uint8, device=dev)  # compute stream: ordered with its readers
                    offs_all, hw_all, base, kept = [], [], 0, []
                    for page in wave["pages"]:
                        path, ids, metas, good, b, at, shape = page
                        nb = int(shape[0]) * int(shape[1]) * int(shape[2])
                        try:
                            _, offs, hw = engine.crop_boxes(pipe["arena"][slot][at : at + nb].view(shape), b, out=pipe["pix"], base=base)
                        except (MmeError, ValueError) as e:
                            logger.error(f"Error in batch processing: {e}")  # this page's regions fail, the wave goes on
                            continue
                        offs_all.append(offs)
                        hw_all.append(hw)
                        base = int(offs[-1]) + (int(hw[-1, 0]) * int(hw[-1, 1]) * 3 + 15) // 16 * 16
                        kept.append(page)
                    wave["pages"] = kept
                    pipe["ev_cut"][slot].record(compute)
                    slot_free[slot].release()  # the producer may refill the arena: its copies wait for ev_cut on the device
                    slot = None
                    tk = tick("consumer: crop_boxes calls", tk)
                    if not kept:
                        continue
                    n = sum(len(o) for o in offs_all)
                    e32, _ = self.embedder.embed_packed(pipe["pix"], np.concatenate(offs_all), np.concatenate(hw_all), want_bf16=False)
                    tk = tick("consumer: embed call", tk)
                except (MmeError, OSError, ValueError, RuntimeError) as e:
                    if slot is not None:
                        pipe["ev_cut"][slot].record(compute)
                        slot_free[slot].release()
                    for path, *_ in wave["pages"]:
                        logger.error(f"Error in batch processing: {e}")  # embedder.py:223-224, once per page the pass carried
                    continue
                rs = n_pass & 1
                n_pass += 1
                pipe["ev_pass"][rs].record(compute)
                if pending is not None and pending[0] == rs:
                    finalize(pending)  # the result buffer is about to be reused
                    pending = None
                if pipe["out"][rs] is None or pipe["out"][rs].shape[0] < n or pipe["out"][rs].shape[1] != e32.shape[1]:
                    pipe["out"][rs] = t.empty((max(n, 2048), e32.shape[1]), dtype=t.float32, pin_memory=cuda)
                with on(pipe["out_stream"]):
                    pipe["out_stream"].wait_event(pipe["ev_pass"][rs])
                    pipe["out"][rs][:n].copy_(e32, non_blocking=True)
                    pipe["ev_out"][rs].record(pipe["out_stream"])
                tk = _time.perf_counter()
                if pending is not None:
                    finalize(pending)  # the previous wave's rows go to the store under this wave's device pass
                tick("consumer: rows of the previous wave to the store", tk)
                pending = (rs, wave, n, e32)
            if pending is not None:
                finalize(pending)
        except BaseException:
            if cuda:
                t.cuda.synchronize(dev)
            raise
        finally:
            stop.set()
            while producer.is_alive():  # drain so that a blocked put() returns
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            producer.join()
        logger.info(f"Completed region processing: {total} regions processed in {total_images} images")
        return total
