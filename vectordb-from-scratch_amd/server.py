"""The production caller of the batched hot path: the reference's `POST /search/batch` endpoint and its metrics
(SURVEY.md 8(f) rank 4), mirrored over the GPU-backed `VectorStore`.

    batch_search   src/server/routes.rs:330-385   one `store.search_batch` (or `_with_filter`) call per request, per-query k
                                                  (default 10), ONE latency sample for the whole batch, errors -> 400
    get_metrics    src/server/routes.rs:417-431   totals + avg / p50 / p95 / p99 query latency in microseconds
    health         src/server/routes.rs:400-415
    MetricsCollector  src/metrics.rs:7-72         counters + an unbounded latency list; percentile = sorted[round(p/100 * (n-1))]
    AppState       src/server/mod.rs:13-16        RwLock<VectorStore>, RwLock<MetricsCollector>: many concurrent searches,
                                                  exclusive writes -- the threading contract of include/vdb_flat.h

JSON shapes are the reference's serde shapes (routes.rs:20-98; `MetadataFilter` is internally tagged: {"op": "eq", "field": ..,
"value": ..}, storage.rs:45-58).  Only the routes on or next to the hot path exist here (insert endpoints to fill a store, the
batch search, health, metrics); the reference's CLI, persistence and the rest of its router stay out of scope.
One extension, off by default: "prefilter": true applies the filter as the device bitmask BEFORE top-k (BASELINE config 4)
instead of the reference's 3k over-fetch post-filter; the reference's result is a prefix of that one.

Starlette is used for routing only (it is in the image); run with `uvicorn vectordb_from_scratch_amd.server:app`-style
factories in a deployment, `starlette.testclient.TestClient` in the tests (no socket is opened, like the reference's
`tower::ServiceExt::oneshot` tests, routes.rs:443, :478).
"""
import threading
import time

from .error import VectorDbError
from .storage import BatchInsertItem, Metadata, MetadataFilter, VectorStore
from .vector import Vector


class MetricsCollector:                                  # src/metrics.rs:7-72
    def __init__(self):
        self.query_latencies_us = []
        self._total_queries = self._total_inserts = self._total_deletes = 0

    def record_query(self, seconds):                     # metrics.rs:25-28 (Duration::as_micros truncates)
        self._total_queries += 1
        self.query_latencies_us.append(float(int(round(seconds * 1e9)) // 1000))     # a Duration holds nanoseconds

    def record_insert(self):
        self._total_inserts += 1

    def record_delete(self):
        self._total_deletes += 1

    def total_queries(self):
        return self._total_queries

    def total_inserts(self):
        return self._total_inserts

    def total_deletes(self):
        return self._total_deletes

    def avg_query_latency_us(self):                      # metrics.rs:52-58
        if not self.query_latencies_us:
            return 0.0
        return sum(self.query_latencies_us) / len(self.query_latencies_us)

    def percentile_query_latency_us(self, percentile):   # metrics.rs:61-71
        if not self.query_latencies_us:
            return 0.0
        s = sorted(self.query_latencies_us)
        x = (percentile / 100.0) * (len(s) - 1)
        index = int(x + 0.5) if x >= 0 else 0            # f64::round: half away from zero
        return s[min(index, len(s) - 1)]


class _RwLock:
    """std::sync::RwLock as the server uses it (server/mod.rs:13-16): shared for searches, exclusive for writes."""

    def __init__(self):
        self._cond = threading.Condition()
        self._readers = 0
        self._writer = False

    def read(self):
        return _Guard(self, False)

    def write(self):
        return _Guard(self, True)


class _Guard:
    def __init__(self, lock, exclusive):
        self.l, self.x = lock, exclusive

    def __enter__(self):
        with self.l._cond:
            if self.x:
                while self.l._writer or self.l._readers:
                    self.l._cond.wait()
                self.l._writer = True
            else:
                while self.l._writer:
                    self.l._cond.wait()
                self.l._readers += 1

    def __exit__(self, *exc):
        with self.l._cond:
            if self.x:
                self.l._writer = False
            else:
                self.l._readers -= 1
            self.l._cond.notify_all()


def filter_from_json(obj):
    """serde(tag = "op", rename_all = "snake_case")  (storage.rs:45-58)."""
    op = obj.get("op")
    if op == "eq":
        return MetadataFilter.Eq(obj["field"], obj["value"])
    if op == "ne":
        return MetadataFilter.Ne(obj["field"], obj["value"])
    if op == "exists":
        return MetadataFilter.Exists(obj["field"])
    if op in ("and", "or"):
        subs = [filter_from_json(f) for f in obj["filters"]]
        return MetadataFilter.And(subs) if op == "and" else MetadataFilter.Or(subs)
    raise ValueError(f"unknown filter op {op!r}")


class AppState:                                           # src/server/mod.rs:13-16
    def __init__(self, store):
        assert isinstance(store, VectorStore)
        self.store, self.metrics = store, MetricsCollector()
        self.store_lock, self.metrics_lock = _RwLock(), _RwLock()


def create_app(state):
    """The router (routes.rs:102-120) restricted to the hot path's callers."""
    from starlette.applications import Starlette
    from starlette.responses import JSONResponse
    from starlette.routing import Route

    def bad_request(msg, status=400):
        return JSONResponse({"error": str(msg)}, status_code=status)

    async def insert_vector(request):                     # routes.rs:124-163 (fills a store for the batch endpoint)
        try:
            req = await request.json()
            item = BatchInsertItem(req["id"], Vector(req["vector"]), Metadata(req.get("metadata") or {}))
        except (KeyError, TypeError, ValueError) as e:
            return bad_request(e, 422)
        try:
            with state.store_lock.write():
                state.store.insert_with_metadata(item.id, item.vector, item.metadata)
        except VectorDbError as e:
            return bad_request(e)
        with state.metrics_lock.write():
            state.metrics.record_insert()
        return JSONResponse({"id": item.id}, status_code=201)

    async def batch_insert(request):                      # routes.rs:283-328
        try:
            req = await request.json()
            items = [BatchInsertItem(v["id"], Vector(v["vector"]), Metadata(v.get("metadata") or {})) for v in req["vectors"]]
        except (KeyError, TypeError, ValueError) as e:
            return bad_request(e, 422)
        try:
            with state.store_lock.write():
                state.store.insert_batch(items)
        except VectorDbError as e:
            return bad_request(e)
        with state.metrics_lock.write():
            for _ in items:
                state.metrics.record_insert()
        return JSONResponse({"inserted": len(items)}, status_code=201)

    def parse_k(v):
        # routes.rs:334-338: `k: Option<usize>`, default 10 -- serde rejects a negative, fractional or non-numeric k (422)
        if v is None:
            return 10
        if isinstance(v, bool) or not isinstance(v, int) or v < 0:
            raise ValueError(f"k must be a non-negative integer, got {v!r}")
        return v

    def run_batch(queries, flt, prefilter):
        with state.store_lock.read():
            # Index::search returns at most len results (flat_index.rs:63): clamp before any buffer is sized by k
            n = state.store.len()
            queries = [(q, min(k, n)) for q, k in queries]
            if flt is not None and prefilter:
                return state.store.search_batch_prefiltered(queries, flt)       # extension: device bitmask before top-k
            if flt is not None:
                return state.store.search_batch_with_filter(queries, flt)       # routes.rs:352-353
            return state.store.search_batch(queries)                            # routes.rs:354-355: ONE index call

    async def batch_search(request):                      # routes.rs:330-385
        from starlette.concurrency import run_in_threadpool
        try:
            req = await request.json()
            queries = [(Vector(q["vector"]), parse_k(q.get("k"))) for q in req["queries"]]
            flt = filter_from_json(req["filter"]) if req.get("filter") is not None else None
        except (KeyError, TypeError, ValueError) as e:
            return bad_request(e, 422)
        start = time.perf_counter()
        try:
            # the blocking GPU call runs off the event loop, so that the shared mode of the RwLock (routes.rs:342) really admits
            # concurrent searches
            all_results = await run_in_threadpool(run_batch, queries, flt, bool(req.get("prefilter")))
        except VectorDbError as e:
            return bad_request(e)
        except (ValueError, OverflowError, MemoryError) as e:
            return bad_request(e)
        elapsed = time.perf_counter() - start
        with state.metrics_lock.write():
            state.metrics.record_query(elapsed)           # one sample for the whole batch (routes.rs:365-369)
        return JSONResponse([[{"id": r.id, "distance": r.distance} for r in res] for res in all_results])

    async def health(request):                            # routes.rs:400-415
        with state.store_lock.read():
            n = state.store.len()
        return JSONResponse({"status": "ok", "vector_count": n})

    async def get_metrics(request):                       # routes.rs:417-431
        with state.metrics_lock.read():
            m = state.metrics
            return JSONResponse({
                "total_queries": m.total_queries(), "total_inserts": m.total_inserts(), "total_deletes": m.total_deletes(),
                "avg_query_latency_us": m.avg_query_latency_us(),
                "p50_query_latency_us": m.percentile_query_latency_us(50.0),
                "p95_query_latency_us": m.percentile_query_latency_us(95.0),
                "p99_query_latency_us": m.percentile_query_latency_us(99.0)})

    return Starlette(routes=[
        Route("/vectors", insert_vector, methods=["POST"]),
        Route("/vectors/batch", batch_insert, methods=["POST"]),
        Route("/search/batch", batch_search, methods=["POST"]),
        Route("/health", health, methods=["GET"]),
        Route("/metrics", get_metrics, methods=["GET"]),
    ])
