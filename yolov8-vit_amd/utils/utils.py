"""Drop-in for the reference's `utils.utils` star-import surface (app.py:8, test.py:2).

Hot path: `Network_Wrapper`, `build_model` (HIP engine underneath).  The remaining names exist
because `app.py` relies on the star import for them (`cv2`, `np`, `os`, `sse`, `torch`, helpers);
they are I/O plumbing outside the hot path and hold NO credentials (the reference hard-codes
some - those literals are deliberately not reproduced; configure through the environment).
"""
import os
import types
import xml.etree.ElementTree as ET

import numpy as np
import torch
import torch.nn as nn  # noqa: F401  (re-exported like the reference module)

try:                                   # OpenCV is optional on the MI355X box: only constants are needed
    import cv2
except Exception:                      # pragma: no cover
    cv2 = types.SimpleNamespace(INTER_NEAREST=0, INTER_LINEAR=1, IMREAD_COLOR=1)
try:
    from flask_sse import sse
except Exception:                      # pragma: no cover
    class _NoSse:
        def publish(self, *a, **k):
            return None
    sse = _NoSse()

from yvhip.modules import Network_Wrapper, build_network  # noqa: E402


def build_model(CFG, modelName=None, pretrained_path=None, pretrained=None):
    """utils/utils.py:75-87 takes `pretrained_path`; app.py:35 and test.py:18 pass `pretrained=`
    (the signature of utils/trainClass.py:341): both spellings are accepted."""
    return build_network(CFG, modelName, pretrained_path or pretrained)


def download_images(url, save_folder, save_flag=True):
    """Fetch one image over HTTP (utils/utils.py:12-56).  Returns the saved path, the decoded array in cv2's BGR
    channel order when save_flag is False (the reference returns cv2.imdecode's result and app.py:71-78 hands it to
    cv2.imwrite), or False on any failure."""
    try:
        import io
        import requests
        from PIL import Image
        resp = requests.get(url, timeout=10)
        resp.raise_for_status()
        img = Image.open(io.BytesIO(resp.content)).convert("RGB")
    except Exception as e:             # network / decode errors are reported, never raised
        print(f"Error downloading {url}: {e}")
        return False
    if not save_flag:
        return np.asarray(img)[..., ::-1].copy()
    name = os.path.basename(url).split("?")[0] or "downloaded_image.jpg"
    os.makedirs(save_folder, exist_ok=True)
    path = os.path.join(save_folder, name)
    try:
        img.save(path)
        return path
    except Exception as e:
        print(f"Error saving image to {path}: {e}")
        return False


class AliyunOss(object):
    """Object-store client placeholder: credentials come from the environment
    (OSS_ACCESS_KEY_ID / OSS_ACCESS_KEY_SECRET / OSS_ENDPOINT / OSS_BUCKET); without them every
    operation reports failure instead of raising, like the reference's error style."""

    def __init__(self):
        self.bucket = None
        self.endpoint = os.environ.get("OSS_ENDPOINT", "")
        self.bucket_name = os.environ.get("OSS_BUCKET", "")
        try:
            import oss2
            kid, sec = os.environ.get("OSS_ACCESS_KEY_ID"), os.environ.get("OSS_ACCESS_KEY_SECRET")
            ep, name = self.endpoint, self.bucket_name
            if kid and sec and ep and name:
                self.bucket = oss2.Bucket(oss2.Auth(kid, sec), ep, name)
        except Exception:
            self.bucket = None

    def put_object_from_file(self, name, file):
        if self.bucket is None:
            return False
        try:
            self.bucket.put_object_from_file(name, file)
            return True
        except Exception as e:
            print(f"oss upload failed: {e}")
            return False

    def getUrl(self, name):
        """Public URL of an object (utils/utils.py:114-116; app.py:101 calls it)."""
        return "https://{}.{}/{}".format(self.bucket_name, self.endpoint, name)

    def delete_object(self, name):
        """utils/utils.py:119-131: True when deleted, False when missing / failing / unconfigured."""
        if self.bucket is None:
            return False
        try:
            self.bucket.delete_object(name)
            return True
        except Exception as e:
            print(f"Error deleting object {name} from OSS: {e}")
            return False


def indent(elem, level=0):
    """Pretty-print helper for ElementTree (two spaces per level)."""
    pad = "\n" + level * "  "
    if len(elem):
        if not elem.text or not elem.text.strip():
            elem.text = pad + "  "
        if not elem.tail or not elem.tail.strip():
            elem.tail = pad
        last = None
        for last in elem:
            indent(last, level + 1)
        if not last.tail or not last.tail.strip():
            last.tail = pad
    elif level and (not elem.tail or not elem.tail.strip()):
        elem.tail = pad


_LABEL_IDS = {'good': '0', 'broke': '1', 'lose': '2', 'loss': '2', 'uncovered': '3', 'circle': '4'}


def generate_annotation(folder_name, image_filename, image_path, objects_data, save_dir="train/new/"):
    """VOC-style XML writer (utils/utils.py:133-228; byte layout pinned by golden G9): <size> is 0/0/3,
    objects use a <sort> tag holding the numeric class id."""
    root = ET.Element("annotation")
    ET.SubElement(root, "folder").text = folder_name
    ET.SubElement(root, "filename").text = image_filename
    ET.SubElement(root, "path").text = image_path
    ET.SubElement(ET.SubElement(root, "source"), "database").text = "Unknown"
    size = ET.SubElement(root, "size")
    ET.SubElement(size, "width").text = "0"
    ET.SubElement(size, "height").text = "0"
    ET.SubElement(size, "depth").text = "3"
    ET.SubElement(root, "segmented").text = "0"
    for obj in objects_data:
        node = ET.SubElement(root, "object")
        sort = obj['sort']
        if isinstance(sort, int):
            text = str(sort)
        elif isinstance(sort, str):
            text = _LABEL_IDS.get(sort, str(sort))
        else:
            text = "unknown"
        ET.SubElement(node, "sort").text = text
        ET.SubElement(node, "pose").text = "Unspecified"
        ET.SubElement(node, "truncated").text = "0"
        ET.SubElement(node, "difficult").text = "0"
        box = ET.SubElement(node, "bndbox")
        for k in ("xmin", "ymin", "xmax", "ymax"):
            ET.SubElement(box, k).text = str(obj[k])
    indent(root)
    os.makedirs(save_dir, exist_ok=True)
    out = os.path.join(save_dir, f"{os.path.splitext(image_filename)[0]}.xml")
    try:
        ET.ElementTree(root).write(out, encoding="utf-8", xml_declaration=False)
        print(f"Annotation XML saved to {out}")
        return out
    except Exception as e:
        print(f"Error writing XML to {out}: {e}")
        return None


def location2lalo(location):
    """Geocode through the Amap REST API (utils/utils.py:248-275); key from AMAP_API_KEY only."""
    key = os.environ.get("AMAP_API_KEY")
    if not key:
        print("AMAP_API_KEY is not set")
        return None, None
    try:
        import requests
        ans = requests.get('https://restapi.amap.com/v3/geocode/geo', params={'address': location, 'key': key},
                           timeout=5).json()
        if ans.get('status') == '1' and ans.get('geocodes'):
            return ans['geocodes'][0]['formatted_address'], ans['geocodes'][0]['location']
        print(f"Error from Amap API: {ans.get('info', 'Unknown error')}")
    except Exception as e:
        print(f"Error requesting Amap API: {e}")
    return None, None


def log(log_queue_obj, message, *args):
    """Queue + SSE log line (utils/utils.py:278-290)."""
    try:
        text = message % args
        if hasattr(log_queue_obj, 'put'):
            log_queue_obj.put(text)
        else:
            print("Warning: log_queue_obj does not have a 'put' method.")
        sse.publish({'message': text}, type='log')
    except Exception as e:
        print(f"Error in log function: {e}")
