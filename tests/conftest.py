import os
import sys
import tarfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# built by the reference with index_documents(map=NULL): our writer reproduces them byte for byte
WRITER_FIXTURES = ["acgt48k", "eng2doc", "counter400_small", "counter400_default", "runs3doc", "construct_kat",
                   "b1000", "bytes256"]
# + one built WITH a document map (buckets carry document chunks, the production femto_index layout)
INDEX_FIXTURES = WRITER_FIXTURES + ["chunks2doc"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Fixture:
    """One committed golden fixture: index files + documents (tar.gz) and reference vectors (npz)."""

    def __init__(self, name, root):
        self.name = name
        self.dir = os.path.join(root, name)
        os.makedirs(self.dir)
        with tarfile.open(os.path.join(GOLDEN, name + ".tar.gz")) as tf:
            tf.extractall(self.dir)
        self.index = os.path.join(self.dir, "index")
        self.flat = os.path.join(self.dir, "index.flat")
        self.gold = np.load(os.path.join(GOLDEN, name + ".npz"))
        docs = sorted(f for f in os.listdir(self.dir) if f.startswith("doc"))
        self.docs = [np.fromfile(os.path.join(self.dir, f), dtype=np.uint8) for f in docs]
        self.doc_paths = [os.path.join(self.dir, f) for f in docs]
        self.params = str(self.gold["params"])

    @property
    def patterns(self):
        plen = self.gold["pat_len"]
        flat = self.gold["pat_flat"]
        starts = np.zeros(len(plen), dtype=np.int64)
        starts[1:] = np.cumsum(plen[:-1])
        return plen.astype(np.int32), flat.astype(np.uint16), starts

    def locate_cases(self):
        for k in self.gold.files:
            if k.startswith("loc") and k.endswith("_noccs"):
                mo = int(k[3:-6])
                yield mo, self.gold[k], self.gold[f"loc{mo}_offs"]

    def prepared_text(self):
        """The reference's prepared text: every document's bytes+5 followed by SEOF (=2)
        (/root/reference/src/main/bwt_prepare.c:227-311)."""
        parts = []
        for d in self.docs:
            parts.append(d.astype(np.uint16) + 5)
            parts.append(np.array([2], dtype=np.uint16))
        return np.concatenate(parts)


@pytest.fixture(scope="session")
def fixtures(tmp_path_factory):
    root = tmp_path_factory.mktemp("golden")
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Fixture(name, str(root))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def gpu_ok():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return True
