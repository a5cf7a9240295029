"""Checkpoint / sample I/O with the same names and behaviour as the
reference's src/utils.py:11-141 (local paths or gs:// URIs, Vertex AI's
AIP_MODEL_DIR).  google-cloud-storage is optional: it is only touched when a
gs:// path is actually used."""
import os
import tempfile
from pathlib import Path
from typing import Tuple, Union

import torch

try:  # optional dependency (absent on the MI355X image)
    from google.cloud import storage  # type: ignore
except Exception:  # pragma: no cover - exercised only without the package
    class _MissingStorage:
        """Stands in for google.cloud.storage; tests patch `.Client`."""

        @staticmethod
        def Client(*_a, **_k):
            raise ImportError("google-cloud-storage is required for gs:// paths")

    storage = _MissingStorage()  # type: ignore

PathLike = Union[str, Path]
_GS = "gs://"


def is_gcs_path(path: PathLike) -> bool:
    """True for gs:// URIs (src/utils.py:11-13)."""
    return str(path).startswith(_GS)


def parse_gcs_path(gcs_path: str) -> Tuple[str, str]:
    """'gs://bucket/a/b' -> ('bucket', 'a/b'); 'gs://bucket' -> ('bucket', '') (src/utils.py:16-24)."""
    if not gcs_path.startswith(_GS):
        raise ValueError(f"Not a GCS path: {gcs_path}")
    bucket, _, blob = gcs_path[len(_GS):].partition("/")
    return bucket, blob


def _blob(gcs_path: str):
    bucket_name, blob_name = parse_gcs_path(gcs_path)
    return storage.Client().bucket(bucket_name).blob(blob_name)


def download_from_gcs(gcs_path: str, local_path: str) -> None:
    """src/utils.py:27-33."""
    _blob(gcs_path).download_to_filename(local_path)


def upload_to_gcs(local_path: str, gcs_path: str) -> None:
    """src/utils.py:36-42."""
    _blob(gcs_path).upload_from_filename(local_path)


def load_checkpoint(ckpt_path: PathLike, device: str) -> dict:
    """torch.load from a local file or via a temp file from GCS (src/utils.py:47-63)."""
    ckpt_path = str(ckpt_path)
    if not is_gcs_path(ckpt_path):
        return torch.load(ckpt_path, map_location=device)
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as tmp:
        try:
            print(f"Downloading checkpoint from GCS: {ckpt_path}")
            download_from_gcs(ckpt_path, tmp.name)
            return torch.load(tmp.name, map_location=device)
        except Exception as e:
            raise RuntimeError(f"Failed to download checkpoint from {ckpt_path}: {e}")
        finally:
            os.unlink(tmp.name)


def save_checkpoint(model_state: dict, ckpt_path: PathLike) -> None:
    """torch.save to a local file or via a temp file to GCS (src/utils.py:66-83)."""
    ckpt_path = str(ckpt_path)
    if not is_gcs_path(ckpt_path):
        torch.save(model_state, ckpt_path)
        print(f"✔ Saved checkpoint to {ckpt_path}")
        return
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as tmp:
        try:
            torch.save(model_state, tmp.name)
            print(f"Uploading checkpoint to GCS: {ckpt_path}")
            upload_to_gcs(tmp.name, ckpt_path)
            print(f"✔ Uploaded checkpoint to {ckpt_path}")
        except Exception as e:
            raise RuntimeError(f"Failed to upload checkpoint to {ckpt_path}: {e}")
        finally:
            os.unlink(tmp.name)


def save_samples(content: Union[str, bytes], sample_path: PathLike, mode: str = "w") -> None:
    """Write text or bytes to a local path (parents created) or GCS (src/utils.py:86-117)."""
    sample_path = str(sample_path)
    if not is_gcs_path(sample_path):
        dest = Path(sample_path)
        dest.parent.mkdir(parents=True, exist_ok=True)
        if isinstance(content, str):
            dest.write_text(content)
        else:
            dest.write_bytes(content)
        print(f"✔ Saved sample to {sample_path}")
        return
    with tempfile.NamedTemporaryFile(mode=mode, suffix=Path(sample_path).suffix, delete=False) as tmp:
        try:
            tmp.write(content)
            tmp.flush()
            tmp.close()  # closed before the upload reads it
            print(f"Uploading sample to GCS: {sample_path}")
            upload_to_gcs(tmp.name, sample_path)
            print(f"✔ Uploaded sample to {sample_path}")
        except Exception as e:
            raise RuntimeError(f"Failed to upload sample to {sample_path}: {e}")
        finally:
            os.unlink(tmp.name)


def get_vertex_checkpoint_path(base_name: str) -> str:
    """AIP_MODEL_DIR/base_name under Vertex AI, else base_name (src/utils.py:120-124)."""
    model_dir = os.environ.get("AIP_MODEL_DIR")
    return os.path.join(model_dir, base_name) if model_dir is not None else base_name


def get_samples_dir(base_dir: str = "samples") -> Union[str, Path]:
    """Samples directory: a str for gs:// model dirs (no Path normalisation), a
    Path otherwise (src/utils.py:127-141)."""
    model_dir = os.environ.get("AIP_MODEL_DIR")
    if model_dir is None:
        return Path(base_dir)
    if model_dir.startswith(_GS):
        return f"{model_dir.rstrip('/')}/{base_dir.strip('/')}"
    return Path(model_dir) / base_dir
