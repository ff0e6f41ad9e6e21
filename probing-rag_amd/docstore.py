"""The CSV docstore that travels with the dense index.

Reference (paths relative to /root/reference):
  * writer  make_indexer.py:461-464  ``pd.DataFrame([texts, doc_ids]).T`` with columns
            ``['doc', 'doc_id']`` -> ``to_csv(..., index=False)``; row r of the CSV is row r of
            the index (both follow the order of ``texts``, make_indexer.py:454-455)
  * reader  exp_rag.py:298           ``corpus = pd.read_csv(...)``
  * lookup  exp_rag.py:436           ``list(corpus.iloc[I[0].tolist(), 0])``
"""


def write_docstore(texts, doc_ids, path: str):
    import pandas as pd
    df = pd.DataFrame([list(texts), list(doc_ids)]).T
    df.columns = ["doc", "doc_id"]
    df.to_csv(path, index=False)
    return df


def read_docstore(path: str):
    import pandas as pd
    return pd.read_csv(path)


def lookup_passages(corpus, ids):
    """exp_rag.py:436: the passages of the retrieved row ids (``I[0].tolist()``).  Like
    ``DataFrame.iloc`` it raises IndexError for ids outside the table and lets -1 (the padding of a
    search with fewer than k rows) wrap to the last row - so callers search with k <= ntotal, as the
    reference does (k = 5)."""
    return list(corpus.iloc[list(ids), 0])


class Docstore:
    """``lookup`` callable for ``retrieve_decide``: ``Docstore(path)(ids) -> list[str]``."""

    def __init__(self, path_or_frame):
        self.corpus = read_docstore(path_or_frame) if isinstance(path_or_frame, str) else path_or_frame

    def __len__(self):
        return len(self.corpus)

    def __call__(self, ids):
        return lookup_passages(self.corpus, ids)
