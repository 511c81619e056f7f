"""COCO result records -- the on-disk format on the far side of the path: what the
reference's eval / test-dev harness writes per image after `Decode.detect_batch` (reference tools/cocotools.py:159-191,
called from :230-247).  Same import name (`from tools.cocotools import ...`), same record layout and rounding:

    {'image_id': id, 'category_id': clsid2catid[class], 'bbox': [xmin, ymin, w, h], 'score': float(score)}
    w = xmax - xmin + 1, h = ymax - ymin + 1        (computed in the dtype of the box array, float32)
    every bbox entry -> round(float(v) * 10) / 10   (Python round: half to even, on the float64 product)

one JSON list per image in `<result_dir>/bbox/<image name without extension>.json`.  The mAP computation itself
(pycocotools) and the drawing (cv2) are host-side tooling outside the path and are not provided.
"""
import json
import os

# contiguous class index -> COCO category id: the ids 1..90 without the ten ids COCO never used
# (the reference spells the table out, tools/cocotools.py:22-38)
_UNUSED_CATIDS = (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)
clsid2catid = {i: c for i, c in enumerate(c for c in range(1, 91) if c not in _UNUSED_CATIDS)}
catid2clsid = {c: i for i, c in clsid2catid.items()}


def bbox_records(boxes, scores, classes, im_id, _clsid2catid=None):
    """Records of one image from the arrays `Decode.detect_image / detect_batch` return (boxes [K,4] xyxy, scores [K],
    classes [K]); reference tools/cocotools.py:168-186."""
    m = clsid2catid if _clsid2catid is None else _clsid2catid
    out = []
    for p in range(len(boxes)):
        xmin, ymin, xmax, ymax = boxes[p]
        w = xmax - xmin + 1
        h = ymax - ymin + 1
        out.append({'image_id': im_id, 'category_id': m[int(classes[p])],
                    'bbox': [round(float(v) * 10) / 10 for v in (xmin, ymin, w, h)], 'score': float(scores[p])})
    return out


def write_bbox_json(result_dir, im_name, boxes, scores, classes, im_id, _clsid2catid=None):
    """One image -> `<result_dir>/bbox/<stem>.json` (reference :187-191).  `boxes is None` (the reference's marker for an
    image without detections in its result lists) writes nothing, like the reference."""
    if boxes is None:
        return None
    path = '%s/bbox/%s.json' % (result_dir, im_name.split('.')[0])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(bbox_records(boxes, scores, classes, im_id, _clsid2catid), f)
    return path


def write_batch(result_dir, result_boxes, result_scores, result_classes, batch_im_id, batch_im_name, _clsid2catid=None):
    """The per-batch loop of the reference's eval (tools/cocotools.py:236-247; its writer threads are an implementation
    detail): `Decode.detect_batch` results -> one file per image.  An image whose detection arrays are empty (this
    build's and the reference's `detect_batch` return three empty arrays for it) gets an empty list, which is what the
    reference writes for `len(boxes) == 0`."""
    paths = []
    for j in range(len(result_boxes)):
        paths.append(write_bbox_json(result_dir, batch_im_name[j], result_boxes[j], result_scores[j], result_classes[j],
                                     batch_im_id[j], _clsid2catid))
    return paths
