"""sqlite document store: the doc-text join after top-k and the helpers the reference keeps in
``inference_pipeline/db_utils/setup_db.py`` (same function names, argument meaning and return shapes).

The reference's driver reads the corpus with ``query(db, "SELECT * FROM documents", fetch_size=50000)``
(``heavy_ranker.py:70-72``) and joins one document per hit with
``query(db, f"SELECT doc FROM documents WHERE id = {uid}", fetch_size=1)`` (``heavy_ranker.py:102-108``).  This
module keeps that API and adds a batched join (one ``IN (...)`` statement per result batch).

Deliberate differences from the reference: where ``setup_db.py`` does ``raise "<str>"`` (a ``TypeError`` in Python 3,
``setup_db.py:23,32,50,68,81,114,127``) this module raises ``sqlite3.OperationalError`` / ``ValueError`` with the same
message text; ``fetch_size`` semantics are unchanged ("all" -> ``fetchall``, N > 1 -> ``fetchmany(N)``, 1 -> ``fetchone``).
"""
from __future__ import annotations

import os
import sqlite3
import threading
from typing import Any, Iterable, List, Sequence, Union

DOCUMENTS_FIELDS = "(id INTEGER PRIMARY KEY AUTOINCREMENT, doc TEXT, source TEXT)"  # setup_db.py:14


def connect_database(database_path: str, verbose: bool = False) -> sqlite3.Connection:
    """``setup_db.py:119-132``: the path must exist and end in ``db`` / ``sqlite``."""
    assert os.path.isfile(database_path), f"Invalid database path for {database_path}"
    assert database_path[-2:] == "db" or database_path[-6:] == "sqlite", \
        "Invalid file, the file must have an extension .db or .sqlite"
    connection = sqlite3.connect(database_path)
    if verbose:
        print(f"Connect to database {database_path} successfully")
    return connection


def setup_database(database_name: str, table_names: List[str] = ("documents",), fields: List[str] = (DOCUMENTS_FIELDS,),
                   database_dir: str = "./inference_pipeline/dbs", verbose: bool = True) -> str:
    """``setup_db.py:12-37``: create ``<database_dir>/<database_name>.db`` with the given tables; returns the path."""
    assert os.path.isdir(database_dir), f"Invalid database_dir path: {database_dir}"
    assert len(table_names) == len(fields), "The table_names and the fields args must have the same length"
    database_path = os.path.join(database_dir, f"{database_name}.db")
    connection = sqlite3.connect(database_path)
    try:
        cursor = connection.cursor()
        for table_name, field in zip(table_names, fields):
            cursor.execute(f"CREATE TABLE IF NOT EXISTS {table_name} {field}")
            if verbose:
                print(f"Successfully create table {table_name} with field {field}")
        connection.commit()
    finally:
        connection.close()
    return database_path


def drop_tables(database_path: str, tables_to_drop: List[str], verbose: bool = True) -> None:
    """``setup_db.py:40-56``."""
    connection = connect_database(database_path, verbose=verbose)
    try:
        cursor = connection.cursor()
        for table_name in tables_to_drop:
            cursor.execute(f"DROP TABLE {table_name}")
        connection.commit()
    finally:
        connection.close()


def query(database_path: str, query_string: str, fetch_size: Union[int, str] = "all", verbose: bool = False) -> Union[list, Any]:
    """``setup_db.py:59-83``: "all" -> list of rows, N > 1 -> up to N rows, 1 -> one row (or ``None``)."""
    connection = connect_database(database_path, verbose=verbose)
    try:
        cursor = connection.cursor()
        cursor.execute(query_string)
        if fetch_size == "all":
            return cursor.fetchall()
        if isinstance(fetch_size, int) and fetch_size > 1:
            return cursor.fetchmany(size=fetch_size)
        if fetch_size == 1:
            return cursor.fetchone()
        raise ValueError("Invalid fetch mode")
    finally:
        connection.close()


def insert_data(database_path: str, table_name: str, data: List[dict], verbose: bool = True) -> None:
    """``setup_db.py:86-116``: bulk insert in one transaction; keys of ``data[0]`` name the columns."""
    if not data:
        return
    connection = connect_database(database_path, verbose=verbose)
    try:
        columns = ", ".join(data[0].keys())
        placeholders = ", ".join(["?"] * len(data[0]))
        values = [tuple(row.values()) for row in data]
        with connection:
            connection.executemany(f"INSERT INTO {table_name} ({columns}) VALUES ({placeholders})", values)
        if verbose:
            print(f"Successfully inserted {len(data)} rows into table {table_name} in {database_path}")
    finally:
        connection.close()


# fetch_docs keeps its connection: opening one costs 120 us, the statement 10 -- and the join runs once per search
# (``heavy_ranker.py:102-108`` runs it once per hit).  One connection per (file, thread), dropped when the file changes.  The cache is
# THREAD-LOCAL (a module-wide dict was mutated without a lock and its flush closed other threads' connections, which sqlite3
# refuses): a thread only ever sees, closes and evicts connections it opened itself, and they go away with the thread.
_read_local = threading.local()


def _read_connection(database_path: str) -> sqlite3.Connection:
    if not os.path.isfile(database_path):
        connect_database(database_path)  # (raises the module's own "no such database" assertion, as before the cache existed)
    st = os.stat(database_path)
    signature = (st.st_ino, st.st_mtime_ns, st.st_size)
    cache = getattr(_read_local, "connections", None)
    if cache is None:
        cache = _read_local.connections = {}
    key = os.path.abspath(database_path)
    hit = cache.get(key)
    if hit is not None and hit[1] == signature:
        return hit[0]
    if hit is not None:
        hit[0].close()
        del cache[key]
    if len(cache) >= 8:  # (a thread that walks over many files: start over -- its own connections only)
        for conn, _ in cache.values():
            conn.close()
        cache.clear()
    connection = connect_database(database_path)
    cache[key] = (connection, signature)
    return connection


def fetch_docs(database_path: str, ids: Sequence[int], table: str = "documents", column: str = "doc") -> dict:
    """Batched doc-text join: ``{id: text}`` for every id of a result batch, one statement per 900 ids
    (the per-hit loop of ``heavy_ranker.py:102-108`` collapsed)."""
    out = {}
    uniq = sorted({int(i) for i in ids if int(i) >= 0})
    if not uniq:
        return out
    cursor = _read_connection(database_path).cursor()
    try:
        for c0 in range(0, len(uniq), 900):
            chunk = uniq[c0:c0 + 900]
            marks = ",".join("?" * len(chunk))
            cursor.execute(f"SELECT id, {column} FROM {table} WHERE id IN ({marks})", chunk)
            out.update({int(r[0]): r[1] for r in cursor.fetchall()})
    finally:
        cursor.close()
    return out


def load_documents(database_path: str, fetch_size: Union[int, str] = 50000, table: str = "documents") -> List[dict]:
    """``heavy_ranker.py:70-76``: rows -> ``[{"id", "text", "source"}]``."""
    rows = query(database_path, f"SELECT * FROM {table}", fetch_size=fetch_size)
    return [{"id": r[0], "text": r[1], "source": r[2]} for r in rows]


def write_documents(database_path: str, documents: Iterable[dict], table: str = "documents") -> None:
    """Create ``table`` (reference schema) if needed and insert ``{"id", "text", "source"}`` rows keeping their ids."""
    directory = os.path.dirname(os.path.abspath(database_path))
    os.makedirs(directory, exist_ok=True)
    connection = sqlite3.connect(database_path)
    try:
        with connection:
            connection.execute(f"CREATE TABLE IF NOT EXISTS {table} {DOCUMENTS_FIELDS}")
            connection.executemany(f"INSERT OR REPLACE INTO {table} (id, doc, source) VALUES (?, ?, ?)",
                                   [(int(d["id"]), d.get("text"), d.get("source")) for d in documents])
    finally:
        connection.close()
