"""Pipeline objects on MI355X -- drop-in for /root/reference/cvpce/production.py
(`ProposalGenerator` :8-20, `Classifier` :22-74, `PlanogramEvaluator` :118-129).

Differences from the reference, all on the device-boundary side:
  * crops / embeddings stay on the GPU (the reference builds crops on the CPU in a Python loop);
  * `BatchedPipeline` runs detect -> crop -> embed -> match for a whole batch of shelf images with a
    single host synchronisation at the end (the reference: one image, one box at a time);
  * degenerate (zero-area after `.to(long)`) boxes make the reference raise inside
    `resize_for_classification`; here they are dropped (documented divergence).
"""
import os

import torch

from . import datautils, ops
from .models.classification import TANH_MEAN, TANH_STD


# Crops / gallery images handed to the encoder per call.  The reference's `batch_size` (8 in cvpce/cli/eval.py, 32 by default) bounds
# the memory of ONE forward pass on its GPU; here a pass over 8 crops would leave 7/8 of the chip idle (a conv5 layer is two
# workgroups per crop), and the per-crop results do not depend on what else is in the batch (no batch statistics; tests assert
# bit-identical embeddings), so `batch_size` is treated as a lower bound and calls are coalesced up to this many images.
ENGINE_BATCH = 1024
INDEX_BATCH = 64      # build_index: gallery images per pass (the host side -- reading and staging f32 images -- dominates there)
PINNED_STAGING = True  # build_index: persistent pinned double buffer instead of torch.stack + pin_memory per pass


def _nondegenerate(boxes):
    b = boxes.to(torch.long)
    return ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)


class ProposalGenerator:
    def __init__(self, detector, device=torch.device('cuda'), confidence_threshold=0.5):
        self.detector = detector
        self.device = device
        self.condfidence_threshold = confidence_threshold  # (sic) production.py:12

    def generate_proposals(self, image):
        res = self.detector(image[None].to(self.device))[0]
        return res['boxes'][res['scores'] > self.condfidence_threshold]

    def _confident_boxes(self, img):
        """The confident, non-degenerate boxes of one image with ONE host synchronisation: the detector's kept boxes are sorted by
        score, so `scores > threshold` is a prefix whose length the post-processing kernel already counted (conf_count); the
        prefix length and the <= detections_per_img boxes come to the host in one small copy, where the degenerate-box test
        (production.py:20 `.to(long)` slices of zero area) costs nothing.  The generic path (boolean masks on the device) costs
        three synchronisations and ~10 tiny launches per image."""
        det = self.detector
        boxes, scores, labels, count, conf, gauss = det.engine().detect([img], det.num_classes, det.detections_per_img, self.condfidence_threshold)
        det.backbone.gaussians = None
        dpi = boxes.shape[1]
        host = self.__dict__.get('_pin')                     # pinned staging buffer: dpi boxes + the prefix length, reused across calls
        if host is None or host.numel() != dpi * 4 + 1:
            host = self._pin = torch.empty(dpi * 4 + 1, dtype=torch.float32).pin_memory()
        packed = torch.cat((boxes[0].reshape(-1), conf[:1].to(torch.float32)))
        host.copy_(packed, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        c = int(host[-1])
        b = host[:c * 4].view(c, 4).to(torch.long)
        ok = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        if bool(ok.all()):
            return boxes[0, :c]
        return boxes[0].index_select(0, ok.nonzero().flatten().to(boxes.device))

    def generate_proposals_and_images(self, image):
        size = datautils.CLASSIFICATION_IMAGE_SIZE
        img = image.to(device=self.device, dtype=torch.float32).contiguous()
        if hasattr(self.detector, 'engine') and not getattr(self.detector, 'training', False):
            boxes = self._confident_boxes(img)
        else:                                   # any other detector object with the reference's call signature
            boxes = self.generate_proposals(image)
            if len(boxes):
                boxes = boxes[_nondegenerate(boxes)]
        if not len(boxes):
            return boxes, torch.empty((0, 3, size, size), device=self.device)
        return boxes, ops.crop_resize(img, boxes, size, mode=0)


class Classifier:
    """production.py:22-74.  `match_dtype`: operand type of the distance GEMM.  float32 (default) is the reference's
    arithmetic (classification.py:87-95 runs in fp32) on the exact-f32 matrix pipe; bfloat16 is the opt-in fast path
    (bench.py / BatchedPipeline pass it explicitly; its top-1 agreement with the fp32 matcher on real embeddings is measured
    by tools/accuracy.py and bounded in tests/test_gpu_accuracy.py)."""

    def __init__(self, encoder, sample_set, device=torch.device('cuda'), emb_device=torch.device('cuda'),
                 batch_size=32, num_workers=8, k=1, load=None, verbose=False, match_dtype=torch.float32):
        self.batch_size = batch_size
        self.num_workers = num_workers  # reader threads of build_index (the reference's DataLoader workers)
        self.device = device
        self.emb_device = emb_device
        self.k = k
        self.encoder = encoder
        self.match_dtype = match_dtype
        if load is None:
            self.embedding, self.annotations = self.build_index(sample_set, verbose)
        else:
            self.embedding, self.annotations = self.load_index(load)
        self._refresh_gallery()

    @classmethod
    def from_embedding(cls, encoder, embedding, annotations, device=torch.device('cuda'), emb_device=torch.device('cuda'),
                       batch_size=32, k=1, match_dtype=torch.float32):
        """Classifier over an index that already exists (e.g. embedded sharded across GPUs and all_gathered)."""
        self = cls.__new__(cls)
        self.batch_size, self.num_workers, self.device, self.emb_device = batch_size, 0, device, emb_device
        self.k, self.encoder, self.match_dtype = k, encoder, match_dtype
        self.set_index(embedding, annotations)
        return self

    def _refresh_gallery(self):
        """Device-resident gallery operand of the distance GEMM (+ its row norms), built once."""
        g = ops.pad_features(self.embedding.to(device=self.device, dtype=self.match_dtype))
        self._gallery = g
        self._gallery_norms = ops.row_norms(g)

    def set_index(self, embedding, annotations):
        self.embedding, self.annotations = embedding, annotations
        self._refresh_gallery()

    def _gallery_items(self, sample_set):
        """sample_set[0], sample_set[1], ... in order, read by `num_workers` threads a bounded window ahead -- the counterpart of
        the reference's DataLoader(num_workers=..., pin_memory=True) (production.py:37-39): decoding and resizing a product
        photo (PIL releases the GIL) would otherwise serialise with the GPU, which embeds > 5 000 gallery images/s."""
        n = len(sample_set)
        if self.num_workers <= 0 or n < 2:
            for i in range(n):
                yield sample_set[i]
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        window = max(2 * self.num_workers, self.batch_size)
        with ThreadPoolExecutor(self.num_workers) as pool:
            pending, nxt = deque(), 0
            while nxt < n or pending:
                while nxt < n and len(pending) < window:
                    pending.append(pool.submit(sample_set.__getitem__, nxt))
                    nxt += 1
                yield pending.popleft().result()

    def build_index(self, sample_set, verbose=False):
        """production.py:36-48.  Gallery images are already in [-1,1] (datautils.py:446): no scale_to_tanh."""
        chunks, annotations, imgs = [], [], []
        step = max(self.batch_size, INDEX_BATCH)
        staging = []          # two pinned host buffers (step, C, H, W) + the event after which each may be refilled

        def flush():
            if not imgs:
                return
            first = imgs[0]
            if first.is_cuda or not torch.cuda.is_available() or not PINNED_STAGING:
                batch = torch.stack(imgs).to(device=self.device)
            else:
                # one pass from the dataset's tensors into a persistent pinned buffer, async upload; the two buffers alternate
                # so that the host fills one while the other's upload / encoder pass is in flight
                if not staging or staging[0][0].shape[1:] != first.shape or staging[0][0].dtype != first.dtype:
                    staging[:] = [[torch.empty((step,) + tuple(first.shape), dtype=first.dtype).pin_memory(), None] for _ in range(2)]
                staging.append(staging.pop(0))
                buf, ev = staging[0]
                if ev is not None:
                    ev.synchronize()
                torch.stack(imgs, out=buf[:len(imgs)])
                batch = buf[:len(imgs)].to(device=self.device, non_blocking=True)
                staging[0][1] = torch.cuda.Event()
                staging[0][1].record()
            chunks.append(self.encoder(batch).detach().to(device=self.emb_device))
            imgs.clear()

        for i, item in enumerate(self._gallery_items(sample_set)):
            imgs.append(item[0])
            annotations.append(item[-1])
            if verbose and (i + 1) % self.batch_size == 0 and (i // self.batch_size) % 100 == 0:
                print(i // self.batch_size)
            if len(imgs) >= step:
                flush()
        flush()
        if chunks:
            embedding = torch.cat(chunks)
        else:
            embedding = torch.empty((0, self.encoder.embedding_size), dtype=torch.float, device=self.emb_device)
        return embedding, annotations

    def save_index(self, pth):
        torch.save({'embedding': self.embedding, 'annotations': self.annotations}, pth)

    def load_index(self, pth):
        idx = torch.load(pth)
        return idx['embedding'], idx['annotations']

    def match(self, emb):
        """(P,1024) f32 unit-norm on device -> (P,k) int64 gallery indices."""
        q = ops.pad_features(emb.to(self.match_dtype))
        return ops.match_topk(q, self._gallery, self.k, g_norms=self._gallery_norms)

    def classify(self, images, return_embedding=False):
        """production.py:57-74: images (P,3,256,256) in [0,1] -> list[list[str]] (P x k)."""
        res, embs = [], []
        eng = self.encoder.engine()
        step = max(self.batch_size, ENGINE_BATCH)
        for i in range(0, len(images), step):
            batch = images[i:i + step].to(device=self.device)
            mean, std = getattr(self.encoder, 'input_mean', TANH_MEAN), getattr(self.encoder, 'input_std', TANH_STD)
            packed = ops.pack_embed_input(batch, True, mean, std)  # scale_to_tanh + the encoder's own normalisation, fused
            if hasattr(eng, 'skip_plan') and batch.dim() == 4 and batch.shape[2] == batch.shape[3] and eng.skip_plan(batch.shape[2]) is not None:
                # crops made by resize_for_classification carry a constant 0.5 border below / right of the box content
                # (datautils.py:232-239): the embedder skips the tiles that lie in it -- the extents are read off the crops
                b32 = batch.to(torch.float32).contiguous()
                emb = eng.embed_packed(packed, ext=ops.pad_extents(b32, 0.5), const_in=eng.const_crop(mean, std, 8, batch.shape[2]))
            else:
                emb = eng.embed_packed(packed)
            if return_embedding:
                embs.append(emb.to(device=self.emb_device))
            nearest = self.match(emb).tolist()
            res += [[self.annotations[j] for j in n] for n in nearest]
        if return_embedding:
            if embs:
                return res, torch.cat(embs)
            return res, torch.empty((0, self.encoder.embedding_size), dtype=torch.float, device=self.emb_device)
        return res


class PlanogramComparator:
    """production.py:76-116: graph matching -> RANSAC homography -> per-label IoU matching -> optional re-classification
    of the expected-but-undetected positions (a second trip through the crop / embed / match kernels)."""

    def __init__(self, graph_threshold=0.5):
        self.graph_threshold = graph_threshold

    def compare(self, expected, actual, image=None, classifier=None):
        from . import planograms
        reproj_threshold = 10 if image is None else min(image.shape[1:]) * 0.01
        if not len(actual['boxes']):
            return 0 if len(expected['boxes']) else 1
        ge = expected['graph'] if 'graph' in expected else planograms.build_graph(expected['boxes'], expected['labels'],
                                                                                  self.graph_threshold)
        ga = planograms.build_graph(actual['boxes'], actual['labels'], self.graph_threshold)
        matching = planograms.large_common_subgraph(ge, ga)
        if not len(matching):
            return 0
        found, missing_indices, missing_positions, missing_labels = planograms.finalize_via_ransac(
            matching, expected['boxes'], actual['boxes'], expected['labels'], actual['labels'],
            reproj_threshold=reproj_threshold)
        if found is None:                       # no homography (production.py:98-99)
            return len(matching) / len(expected['boxes'])
        if classifier is not None and image is not None and len(missing_positions):
            h, w = image.shape[1:]
            pos = missing_positions.clone()
            pos[:, 0::2] = pos[:, 0::2].clamp(min=0, max=w)
            pos[:, 1::2] = pos[:, 1::2].clamp(min=0, max=h)
            valid = (pos[:, 2] - pos[:, 0] > 1) & (pos[:, 3] - pos[:, 1] > 1)
            if not valid.any():
                return found.sum() / len(found)
            idx, pos = missing_indices[valid], pos[valid]
            lbls = [l for l, v in zip(missing_labels, valid) if v]
            img = image.to(device=classifier.device, dtype=torch.float32).contiguous()
            crops = ops.crop_resize(img, pos.to(classifier.device), datautils.CLASSIFICATION_IMAGE_SIZE, mode=0)
            for i, want, got in zip(idx, lbls, classifier.classify(crops)):
                if want == got[0]:
                    found[i] = True
        return found.sum() / len(found)


class PlanogramEvaluator:
    """production.py:118-129 glue; `planogram_comparator` is any object with `.compare(expected, actual, image, classifier)`."""

    def __init__(self, proposal_generator, classifier, planogram_comparator):
        self.proposal_generator = proposal_generator
        self.classifier = classifier
        self.planogram_comparator = planogram_comparator

    def evaluate(self, image, planogram):
        boxes, images = self.proposal_generator.generate_proposals_and_images(image)
        classes = [ann[0] for ann in self.classifier.classify(images)]
        return self.planogram_comparator.compare(planogram, {'boxes': boxes.detach().cpu(), 'labels': classes},
                                                 image, self.classifier)

    def detect_and_classify_batch(self, images):
        """[(boxes (P,4) cpu, labels list[str])] for a list of images -- what `evaluate` hands to the comparator, per image, but
        computed for the whole list at once: images of one size go through `BatchedPipeline` together (one detector pass, one
        embedder schedule, one distance GEMM).  Per-image results are the ones `evaluate` gets: neither the detector's nor the
        embedder's result for an image depends on what it is batched with (tests/test_gpu_harness.py), images of different
        sizes are never mixed in a batch (the padded batch shape enters the anchors, like in torchvision), and the zero-area
        boxes `generate_proposals_and_images` drops are dropped here too."""
        pg = self.proposal_generator
        pipe = self.__dict__.get('_pipe')
        if pipe is None or pipe.detector is not pg.detector or pipe.classifier is not self.classifier:
            pipe = self._pipe = BatchedPipeline(pg.detector, self.classifier, pg.condfidence_threshold)
        pipe.confidence_threshold = pg.condfidence_threshold
        out = [None] * len(images)
        groups = {}
        for i, img in enumerate(images):
            groups.setdefault(tuple(img.shape), []).append(i)
        for idxs in groups.values():
            res = pipe.run([images[i].to(device=pg.device, dtype=torch.float32).contiguous() for i in idxs])
            boxes, indices = res['boxes'].cpu(), res['indices'][:, :, 0].cpu()
            for j, i in enumerate(idxs):
                c = res['counts_host'][j]
                b, ix = boxes[j, :c], indices[j, :c]
                if c:
                    ok = _nondegenerate(b)
                    b, ix = b[ok], ix[ok]
                out[i] = (b, [self.classifier.annotations[k] for k in ix.tolist()])
        return out

    def evaluate_batch(self, images, planograms):
        """[evaluate(image, planogram) for image, planogram in zip(...)] with the detect / crop / embed / match half batched
        (`detect_and_classify_batch`); the comparator (CPU graph matching, RANSAC, re-classification of missing positions) runs
        per image as in production.py:123-129."""
        dets = self.detect_and_classify_batch(images)
        return [self.planogram_comparator.compare(plano, {'boxes': b, 'labels': labels}, img, self.classifier)
                for img, plano, (b, labels) in zip(images, planograms, dets)]


    def evaluate_iter(self, pairs, lookahead=4):
        """`evaluate(image, planogram)` over an ITERABLE of (image, planogram) pairs, yielding the results in order -- the per-image
        calling pattern of production.py:118-129 (one shelf photo at a time, e.g. a camera feed or a dataset loop) without its cost:
        a single image is a 1.9 ms latency chain through the detector and 200 crops do not fill the embedder's passes (171 images/s
        against 275 batched).  The iterator reads up to `lookahead` pairs ahead and sends consecutive images of one size through
        `detect_and_classify_batch` together; the comparator runs per image as before.  Every yielded value equals what
        `evaluate(image, planogram)` returns for that pair (tests/test_gpu_harness.py): an image's detections and labels do not depend
        on what it is batched with.  lookahead = 1 is the serial loop."""
        window = []

        def flush():
            dets = self.detect_and_classify_batch([img for img, _ in window])
            for (img, plano), (b, labels) in zip(window, dets):
                yield self.planogram_comparator.compare(plano, {'boxes': b, 'labels': labels}, img, self.classifier)
            window.clear()

        for image, planogram in pairs:
            if window and (len(window) >= max(1, lookahead) or tuple(image.shape) != tuple(window[0][0].shape)):
                yield from flush()
            window.append((image, planogram))
        if window:
            yield from flush()


CROP_CONTENT_ONLY = os.environ.get('CVPCE_CROP_CONTENT', '1') != '0'   # A/B switch: the crop kernel leaves the constant padding unwritten for the work-list embedder


class BatchedPipeline:
    """detect -> RoI crop -> embed -> match for a batch of shelf images, device-resident end to end.

    Same arithmetic as ProposalGenerator.generate_proposals_and_images + Classifier.classify per image
    (production.py:13-20,57-74) but: one detector pass for the whole batch, the crop kernel reads the
    kept boxes and the confidence-prefix count straight from device memory and writes normalised
    NHWC bf16 embedder input (no f32 crop tensor, no host round trip), one distance GEMM per batch.
    """

    def __init__(self, detector, classifier, confidence_threshold=0.5):
        self.detector = detector
        self.classifier = classifier
        self.confidence_threshold = confidence_threshold

    def _crops(self, images, det_out, content_only=True):
        """RoI crops of every image's confident boxes, launched WITHOUT knowing the counts on the host: the crop kernel reads
        the boxes and the confidence-prefix count from device memory and skips the slots beyond it.
        -> (crops, their content extents | None, the encoder's all-padding crop | None, partial).
        partial (content_only and an encoder that runs the work-list schedule): only the crops' content was written -- the constant
        padding beyond the extents is never read by that schedule (it reads the constant crop instead), so it is not produced either;
        `embed_packed(..., partial=True)` refuses to run any other schedule on such crops."""
        boxes, scores, labels, count, conf_count, gauss = det_out
        eng = self.detector.engine()
        emb_eng = self.classifier.encoder.engine()
        dpi = self.detector.detections_per_img
        size = datautils.CLASSIFICATION_IMAGE_SIZE
        n = len(images)
        # the fused VGG stem reads 4- or 8-channel pixels: 8-byte pixels halve what the crop kernel writes and the stem re-reads
        narrow = getattr(emb_eng, 'stem', None) is not None
        mean, std = getattr(self.classifier.encoder, 'input_mean', TANH_MEAN), getattr(self.classifier.encoder, 'input_std', TANH_STD)
        crops = torch.empty((n * dpi, size, size, 4 if narrow else 8), dtype=torch.bfloat16, device=eng.device)
        # content extents of the crops (the part of a crop beyond them is the pad constant of datautils.py:232-239: the embedder
        # skips the tiles that lie in it) -- only for an encoder whose engine has the work-list schedule
        skip = hasattr(emb_eng, 'skip_plan') and emb_eng.skip_plan(size) is not None
        ext = torch.empty((n * dpi, 2), dtype=torch.int32, device=eng.device) if skip else None
        same_size = all(img.shape == images[0].shape for img in images)
        partial = bool(content_only and skip and CROP_CONTENT_ONLY and emb_eng.will_skip(size))
        if skip and same_size:                   # (images of one size: the extents of the whole batch in one launch)
            ops.crop_extents(boxes.reshape(n * dpi, 4), conf_count, images[0].shape[1], images[0].shape[2], size, out=ext, per_image=dpi)
        for i, img in enumerate(images):
            if skip and not same_size:
                ops.crop_extents(boxes[i], conf_count[i:i + 1], img.shape[1], img.shape[2], size, out=ext[i * dpi:(i + 1) * dpi])
            ops.crop_resize(img, boxes[i], size, mode=2 if narrow else 1, mean=mean, std=std, count=conf_count[i:i + 1],
                            out=crops[i * dpi:(i + 1) * dpi], content_ext=ext[i * dpi:(i + 1) * dpi] if partial else None)
        return crops, ext, (emb_eng.const_crop(mean, std, crops.shape[3], size) if skip else None), partial

    def _select(self, crops, counts, ext=None):
        """The embedder only runs over the valid crops (compaction = a gather of row indices; their extents go along)."""
        n, dpi = len(counts), self.detector.detections_per_img
        eng = self.detector.engine()
        if sum(counts) == n * dpi:
            valid, sel = crops, None
        else:
            sel = torch.cat([torch.arange(i * dpi, i * dpi + c, device=eng.device) for i, c in enumerate(counts)])
            valid = crops.index_select(0, sel)
            if ext is not None:
                ext = ext.index_select(0, sel)
        return valid, sel, ext

    def _crop_embed_match(self, images, det_out, counts, embed_batch=None):
        """(kept for the dev tools) crops + selection with counts already on the host."""
        crops, ext, const_in, _ = self._crops(images, det_out, content_only=False)     # (whole crops: the tools embed them without lists too)
        valid, sel, ext = self._select(crops, counts, ext)
        return crops, valid, sel

    def _finish(self, images, det_out, counts, emb, idx, sel):
        boxes, scores, labels, count, conf_count, gauss = det_out
        n, dpi = len(images), self.detector.detections_per_img
        k = idx.shape[1] if idx.numel() else self.classifier.k
        indices = torch.full((n * dpi, k), -1, dtype=torch.int64, device=boxes.device)
        if sel is None:
            indices = idx
        elif idx.numel():
            indices.index_copy_(0, sel, idx)
        return {'boxes': boxes, 'scores': scores, 'labels': labels, 'count': conf_count, 'det_count': count,
                'indices': indices.view(n, dpi, k), 'gaussians': gauss, 'embeddings': emb, 'counts_host': counts}

    @torch.no_grad()
    def run(self, images, stage_events=None, proposals=None):
        """images: list of (3,H,W) f32 cuda tensors -> dict of device tensors:
        boxes (N,dpi,4), scores (N,dpi), count (N,) = #scores > confidence, indices (N,dpi,k) (-1 beyond count).
        stage_events: optional list that receives (stage name, start event, end event) for detect / crop / embed / match.
        proposals: optional (boxes (N,dpi,4) f32, counts (N,) int32) device tensors that REPLACE the detector's confident boxes after
        the detector has run (stage-isolated benchmarks accept planted proposals, which pins P and the box shapes
        whatever the detector's weights emit); the detector's pass is still executed and timed.
        (Queueing the detector of call i + 1 on its own stream beside the embedder of call i was measured: +1.3 % at best, and
        the two stages' queues do not actually make progress side by side on this stack -- profiles/r03_rejected_experiments.md.)"""
        def mark():
            if stage_events is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e

        det = self.detector
        t0 = mark()
        if ops.PROFILE is not None:
            ops.PROFILE.records.stage = 'detect'           # (per-launch profile: the detector's 3x3 halo launches are filed apart)
        det_out = det.engine().detect(images, det.num_classes, det.detections_per_img, self.confidence_threshold)
        if ops.PROFILE is not None:
            ops.PROFILE.records.stage = 'embed'
        if proposals is not None:
            pb, pc = proposals
            assert pb.shape == det_out[0].shape and pb.dtype == torch.float32 and pc.dtype == torch.int32 and pc.shape == det_out[4].shape
            det_out = (pb, det_out[1], det_out[2], det_out[3], pc, det_out[5])
        t1 = mark()
        # The one host synchronisation of a step -- the confidence-prefix counts, which size the embedder's batch -- is
        # taken BESIDE the crop kernels, not before them: the counts go to pinned host memory right behind the detector,
        # the crops (which read the counts on the device) are launched behind that copy, and the host wakes up as soon as
        # the copy has landed, i.e. while the crops still run, and queues the embedder's launches behind them.
        n = len(images)
        pin = self.__dict__.setdefault('_count_pins', {})
        if n not in pin:
            pin[n] = torch.empty(n, dtype=torch.int32).pin_memory()
        pin[n].copy_(det_out[4], non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        crops, ext, const_in, partial = self._crops(images, det_out)
        copied.synchronize()
        counts = pin[n].tolist()
        valid, sel, ext = self._select(crops, counts, ext)
        t2 = mark()
        if ext is not None:         # (an encoder with the work-list schedule: the tiles in the crops' constant padding are skipped)
            emb = self.classifier.encoder.engine().embed_packed(valid, ext=ext, const_in=const_in, partial=partial)
        else:
            emb = self.classifier.encoder.engine().embed_packed(valid)
        t3 = mark()
        idx = self.classifier.match(emb)
        t4 = mark()
        if stage_events is not None:
            stage_events += [('detect', t0, t1), ('crop', t1, t2), ('embed', t2, t3), ('match', t3, t4)]
        return self._finish(images, det_out, counts, emb, idx, sel)
