// bbq_persist.cpp - on-disk format: <prefix>.vemb (metadata) + <prefix>.veb (the device tile records, byte for byte)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include "bbq_host.h"

using namespace bbq;

namespace {

#pragma pack(push, 1)
struct MetaHeader {
  char magic[4];  // "BVEC" (COMPONENT_NAMES.BINARIZED_VECTOR, src/constants.ts:62-65)
  uint32_t version;
  // MetadataFormat, src/types.ts:92-113
  int32_t fieldNumber, vectorEncodingOrdinal, vectorSimilarityOrdinal, dimensions;
  int64_t vectorDataOffset, vectorDataLength, vectorCount;
  double centroidSquareMagnitude;
  // geometry of the tile records in the vector-data file (= the device layout, bbq_device.h)
  int32_t indexBits, layout, w16, tileStride, hasX1, tileRows;
  int64_t tilesBytes, exactBytes, rowBase;
};
#pragma pack(pop)
static_assert(sizeof(MetaHeader) == 104, "vemb header layout");
// version 3 = version 2 + the pilot replica of a row shard (bbq_index_create_shard): three more words behind the header and the replica's
// tile records + side section behind the shard's own in the data file.  Written only for a shard that has one; version 2 otherwise.
struct PilotExt {
  int64_t pilotRows, pilotTilesBytes, pilotExactBytes;
};
constexpr uint32_t kFileVersion = 2;  // 2: compact tiles carry 4 B per row; the side section = exact corrections + per-tile add ranges
constexpr uint32_t kFileVersionPilot = 3;

// manifest of a multi-device index: <prefix>.vemb with this header, then {rowBase, rows} per shard, the centroid, one checksum;
// shard s lives in the ordinary pair <prefix>.s<NNN>.veb/.vemb (version 3 when it carries a pilot replica)
#pragma pack(push, 1)
struct MultiHeader {
  char magic[4];  // "BVEM"
  uint32_t version;
  int32_t nShards, dimensions, indexBits, vectorSimilarityOrdinal;
  int64_t vectorCount;
  double centroidSquareMagnitude;
  int64_t pilotRows;
};
#pragma pack(pop)
static_assert(sizeof(MultiHeader) == 48, "multi manifest layout");
constexpr uint32_t kMultiVersion = 1;

// FNV-1a over little-endian 64-bit words (tail zero-padded): one multiply per 8 bytes keeps up with the disk
uint64_t fnv64_words(const void *data, size_t n, uint64_t h) {
  const uint8_t *p = (const uint8_t *)data;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
  }
  if (i < n) {
    uint64_t w = 0;
    memcpy(&w, p + i, n - i);
    h = (h ^ w) * 0x100000001b3ull;
  }
  return h;
}
constexpr uint64_t kFnvSeed = 0xcbf29ce484222325ull;

int32_t expected_tile_stride(int32_t w16, int32_t layout, int32_t has_x1) {
  return tile_stride_of(w16, layout, has_x1);
}

struct FileCloser {
  FILE *f;
  ~FileCloser() { if (f) fclose(f); }
};

int read_meta(const char *prefix, MetaHeader *h, std::vector<float> *centroid, uint64_t *data_sum, PilotExt *pilot = nullptr) {
  const std::string path = std::string(prefix) + ".vemb";
  FileCloser fc{fopen(path.c_str(), "rb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot open %s", path.c_str());
  if (fread(h, sizeof *h, 1, fc.f) != 1) return fail(BBQ_ERR_INVALID_ARG, "%s: truncated header", path.c_str());
  if (memcmp(h->magic, "BVEC", 4) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: not a BVEC metadata file", path.c_str());
  if (memcmp(h->magic, "BVEC", 4) == 0 && h->version != kFileVersion && h->version != kFileVersionPilot)
    return fail(BBQ_ERR_UNSUPPORTED, "%s: format version %u (this build reads %u and %u)", path.c_str(), h->version, kFileVersion, kFileVersionPilot);
  PilotExt px{0, 0, 0};
  if (h->version == kFileVersionPilot && fread(&px, sizeof px, 1, fc.f) != 1) return fail(BBQ_ERR_INVALID_ARG, "%s: truncated header", path.c_str());
  if (h->dimensions <= 0 || h->indexBits < 1 || h->indexBits > 8 || !dim_supported(h->dimensions, h->dimensions == 1 ? 1 : store_bits_of(h->indexBits)) || h->vectorCount < 0 || h->rowBase < 0 || h->indexBits < 1 || h->indexBits > 8 || h->tileRows != kTileRows ||
      (h->layout != kLayoutCompact && h->layout != kLayoutInline) || (h->hasX1 != 0 && h->hasX1 != 1) ||
      h->vectorSimilarityOrdinal < 0 || h->vectorSimilarityOrdinal > 2)
    return fail(BBQ_ERR_INVALID_ARG, "%s: header fields out of range", path.c_str());
  const int32_t pb = row_bytes_of(h->dimensions, h->dimensions == 1 ? 1 : store_bits_of(h->indexBits));
  const int64_t n_tiles = (h->vectorCount + kTileRows - 1) / kTileRows;
  if (h->w16 != (pb + 15) / 16 || h->tileStride != expected_tile_stride(h->w16, h->layout, h->hasX1) ||
      (h->layout == kLayoutCompact && h->hasX1) || h->tilesBytes != n_tiles * h->tileStride ||
      h->exactBytes != (h->layout == kLayoutCompact ? compact_side_bytes(n_tiles) : 0) ||
      h->vectorDataLength != h->tilesBytes + h->exactBytes + px.pilotTilesBytes + px.pilotExactBytes || h->vectorDataOffset < 0)
    return fail(BBQ_ERR_INVALID_ARG, "%s: tile geometry does not match dimensions/vectorCount", path.c_str());
  if (h->version == kFileVersionPilot) {
    const int64_t p_tiles = (px.pilotRows + kTileRows - 1) / kTileRows;
    if (px.pilotRows <= 0 || px.pilotRows > h->rowBase || px.pilotTilesBytes != p_tiles * h->tileStride ||
        px.pilotExactBytes != (h->layout == kLayoutCompact ? compact_side_bytes(p_tiles) : 0))
      return fail(BBQ_ERR_INVALID_ARG, "%s: pilot replica geometry does not match", path.c_str());
  }
  std::vector<float> cen((size_t)h->dimensions);
  uint64_t sums[2];
  if (fread(cen.data(), 4, cen.size(), fc.f) != cen.size() || fread(sums, 8, 2, fc.f) != 2)
    return fail(BBQ_ERR_INVALID_ARG, "%s: truncated", path.c_str());
  uint64_t m = fnv64_words(h, sizeof *h, kFnvSeed);
  if (h->version == kFileVersionPilot) m = fnv64_words(&px, sizeof px, m);
  m = fnv64_words(cen.data(), cen.size() * 4, m);
  m = fnv64_words(&sums[0], 8, m);
  if (m != sums[1]) return fail(BBQ_ERR_INVALID_ARG, "%s: metadata checksum mismatch", path.c_str());
  if (pilot) *pilot = px;
  if (centroid) centroid->swap(cen);
  if (data_sum) *data_sum = sums[0];
  return BBQ_OK;
}

// the manifest of a multi-device index (read without a device)
struct Manifest {
  MultiHeader h{};
  std::vector<int64_t> bounds;  // {rowBase, rows} per shard
  std::vector<float> centroid;
};

bool is_manifest(const char *prefix) {
  const std::string path = std::string(prefix) + ".vemb";
  FileCloser fc{fopen(path.c_str(), "rb")};
  char magic[4] = {0, 0, 0, 0};
  return fc.f && fread(magic, 1, 4, fc.f) == 4 && memcmp(magic, "BVEM", 4) == 0;
}

int read_manifest(const char *prefix, Manifest *m) {
  const std::string path = std::string(prefix) + ".vemb";
  FileCloser fc{fopen(path.c_str(), "rb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot open %s", path.c_str());
  MultiHeader &h = m->h;
  if (fread(&h, sizeof h, 1, fc.f) != 1 || memcmp(h.magic, "BVEM", 4) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: not a multi-device manifest", path.c_str());
  if (h.version != kMultiVersion) return fail(BBQ_ERR_UNSUPPORTED, "%s: manifest version %u (this build reads %u)", path.c_str(), h.version, kMultiVersion);
  if (h.nShards < 1 || h.nShards > 64 || h.dimensions <= 0 || h.indexBits < 1 || h.indexBits > 8 || h.vectorCount < 0 || h.pilotRows < 0 ||
      h.vectorSimilarityOrdinal < 0 || h.vectorSimilarityOrdinal > 2 || !dim_supported(h.dimensions, h.dimensions == 1 ? 1 : store_bits_of(h.indexBits)))
    return fail(BBQ_ERR_INVALID_ARG, "%s: manifest fields out of range", path.c_str());
  m->bounds.resize((size_t)h.nShards * 2);
  m->centroid.resize((size_t)h.dimensions);
  uint64_t sum = 0;
  if (fread(m->bounds.data(), 8, m->bounds.size(), fc.f) != m->bounds.size() || fread(m->centroid.data(), 4, m->centroid.size(), fc.f) != m->centroid.size() ||
      fread(&sum, 8, 1, fc.f) != 1)
    return fail(BBQ_ERR_INVALID_ARG, "%s: truncated", path.c_str());
  uint64_t c = fnv64_words(&h, sizeof h, kFnvSeed);
  c = fnv64_words(m->bounds.data(), m->bounds.size() * 8, c);
  c = fnv64_words(m->centroid.data(), m->centroid.size() * 4, c);
  if (c != sum) return fail(BBQ_ERR_INVALID_ARG, "%s: manifest checksum mismatch", path.c_str());
  int64_t at = 0;
  for (int s = 0; s < h.nShards; ++s) {  // contiguous, ascending, covering [0, vectorCount)
    if (m->bounds[(size_t)2 * s] != at || m->bounds[(size_t)2 * s + 1] < 0) return fail(BBQ_ERR_INVALID_ARG, "%s: shards are not contiguous", path.c_str());
    at += m->bounds[(size_t)2 * s + 1];
  }
  if (at != h.vectorCount) return fail(BBQ_ERR_INVALID_ARG, "%s: shards do not add up to vectorCount", path.c_str());
  return BBQ_OK;
}

std::string shard_prefix(const char *prefix, int s) {
  char buf[32];
  snprintf(buf, sizeof buf, ".s%03d", s);
  return std::string(prefix) + buf;
}

}  // namespace

namespace bbq {

// <prefix>.vemb of a multi-device index; the shards' own pairs are written by bbq_index_save on each shard handle
int write_manifest(const char *prefix, int32_t n_shards, const int64_t *bounds, int32_t dim, int32_t index_bits, int32_t sim, int64_t n_rows,
                   double centroid_dp, int64_t pilot_rows, const float *centroid) {
  MultiHeader h{};
  memcpy(h.magic, "BVEM", 4);
  h.version = kMultiVersion;
  h.nShards = n_shards;
  h.dimensions = dim;
  h.indexBits = index_bits;
  h.vectorSimilarityOrdinal = sim;
  h.vectorCount = n_rows;
  h.centroidSquareMagnitude = centroid_dp;
  h.pilotRows = pilot_rows;
  uint64_t c = fnv64_words(&h, sizeof h, kFnvSeed);
  c = fnv64_words(bounds, (size_t)n_shards * 16, c);
  c = fnv64_words(centroid, (size_t)dim * 4, c);
  const std::string path = std::string(prefix) + ".vemb";
  FileCloser fc{fopen(path.c_str(), "wb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot create %s", path.c_str());
  if (fwrite(&h, sizeof h, 1, fc.f) != 1 || fwrite(bounds, 16, (size_t)n_shards, fc.f) != (size_t)n_shards ||
      fwrite(centroid, 4, (size_t)dim, fc.f) != (size_t)dim || fwrite(&c, 8, 1, fc.f) != 1 || fflush(fc.f) != 0)
    return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", path.c_str());
  return BBQ_OK;
}

std::string shard_file_prefix(const char *prefix, int s) { return shard_prefix(prefix, s); }

}  // namespace bbq

extern "C" {

int bbq_index_save(bbq_index *ix, const char *prefix, const float *centroid, int32_t sim) {
  clear_error();
  if (!ix || !prefix || !centroid) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_save: null argument");
  if (sim < 0 || sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (ix->multi) return multi_save(ix, prefix, centroid, sim);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  int rcs = settle_shard_slots(ix->ctx, ix);
  if (rcs != BBQ_OK) return rcs;
  const int64_t n_tiles = (ix->n_rows + kTileRows - 1) / kTileRows;
  const int64_t p_rows = ix->has_pilot ? ix->pilot.view.n_rows : 0, p_tiles = (p_rows + kTileRows - 1) / kTileRows;
  MetaHeader h{};
  memcpy(h.magic, "BVEC", 4);
  h.version = ix->has_pilot ? kFileVersionPilot : kFileVersion;
  h.vectorSimilarityOrdinal = sim;
  h.dimensions = ix->dim;
  h.vectorCount = ix->n_rows;
  h.centroidSquareMagnitude = ix->centroid_dp;
  h.indexBits = ix->index_bits;
  h.layout = ix->layout;
  h.w16 = ix->w16;
  h.tileStride = ix->tile_stride;
  h.hasX1 = ix->has_x1;
  h.tileRows = kTileRows;
  h.tilesBytes = n_tiles * ix->tile_stride;
  h.exactBytes = ix->layout == kLayoutCompact ? compact_side_bytes(n_tiles) : 0;
  h.rowBase = ix->row_base;
  PilotExt px{p_rows, p_tiles * ix->tile_stride, (ix->has_pilot && ix->layout == kLayoutCompact) ? compact_side_bytes(p_tiles) : 0};
  h.vectorDataOffset = 0;
  h.vectorDataLength = h.tilesBytes + h.exactBytes + px.pilotTilesBytes + px.pilotExactBytes;
  const std::string dpath = std::string(prefix) + ".veb", mpath = std::string(prefix) + ".vemb";
  uint64_t dsum = kFnvSeed;
  {
    FileCloser fc{fopen(dpath.c_str(), "wb")};
    if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot create %s", dpath.c_str());
    const size_t piece = 64u << 20;  // multiple of 8: the checksum words never straddle pieces
    std::vector<uint8_t> buf(piece);
    const uint8_t *src[4] = {ix->main.d_tiles, (const uint8_t *)ix->main.d_exact, ix->pilot.d_tiles, (const uint8_t *)ix->pilot.d_exact};
    const int64_t len[4] = {h.tilesBytes, h.exactBytes, px.pilotTilesBytes, px.pilotExactBytes};
    for (int part = 0; part < 4; ++part) {
      for (int64_t o = 0; o < len[part]; o += (int64_t)piece) {
        const size_t m = (size_t)std::min<int64_t>((int64_t)piece, len[part] - o);
        HIPCHK(hipMemcpy(buf.data(), src[part] + o, m, hipMemcpyDeviceToHost));
        dsum = fnv64_words(buf.data(), m, dsum);
        if (fwrite(buf.data(), 1, m, fc.f) != m) return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", dpath.c_str());
      }
    }
    if (fflush(fc.f) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", dpath.c_str());
  }
  FileCloser fc{fopen(mpath.c_str(), "wb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot create %s", mpath.c_str());
  uint64_t m = fnv64_words(&h, sizeof h, kFnvSeed);
  if (ix->has_pilot) m = fnv64_words(&px, sizeof px, m);
  m = fnv64_words(centroid, (size_t)ix->dim * 4, m);
  m = fnv64_words(&dsum, 8, m);
  const uint64_t sums[2] = {dsum, m};
  if (fwrite(&h, sizeof h, 1, fc.f) != 1 || (ix->has_pilot && fwrite(&px, sizeof px, 1, fc.f) != 1) ||
      fwrite(centroid, 4, (size_t)ix->dim, fc.f) != (size_t)ix->dim || fwrite(sums, 8, 2, fc.f) != 2 || fflush(fc.f) != 0)
    return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", mpath.c_str());
  return BBQ_OK;
}

int bbq_index_file_info(const char *prefix, int64_t *n_rows, int32_t *dim, int32_t *sim, double *cdp, int64_t *row_base) {
  clear_error();
  if (!prefix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_file_info: null path");
  if (is_manifest(prefix)) {  // a multi-device index: the totals
    Manifest m;
    int rc = read_manifest(prefix, &m);
    if (rc != BBQ_OK) return rc;
    if (n_rows) *n_rows = m.h.vectorCount;
    if (dim) *dim = m.h.dimensions;
    if (sim) *sim = m.h.vectorSimilarityOrdinal;
    if (cdp) *cdp = m.h.centroidSquareMagnitude;
    if (row_base) *row_base = 0;
    return BBQ_OK;
  }
  MetaHeader h;
  int rc = read_meta(prefix, &h, nullptr, nullptr);
  if (rc != BBQ_OK) return rc;
  if (n_rows) *n_rows = h.vectorCount;
  if (dim) *dim = h.dimensions;
  if (sim) *sim = h.vectorSimilarityOrdinal;
  if (cdp) *cdp = h.centroidSquareMagnitude;
  if (row_base) *row_base = h.rowBase;
  return BBQ_OK;
}

int32_t bbq_index_file_shards(const char *prefix) {
  clear_error();
  if (!prefix) return 0;
  if (!is_manifest(prefix)) {
    MetaHeader h;
    return read_meta(prefix, &h, nullptr, nullptr) == BBQ_OK ? 1 : 0;
  }
  Manifest m;
  return read_manifest(prefix, &m) == BBQ_OK ? m.h.nShards : 0;
}

int bbq_index_load_multi(const char *prefix, int32_t n_devices, const int32_t *devices, bbq_index **out, float *centroid_out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load_multi: out is null");
  *out = nullptr;
  if (!prefix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load_multi: null path");
  if (n_devices < 0 || (n_devices > 0 && !devices)) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load_multi: bad device list");
  Manifest m;
  int rc = read_manifest(prefix, &m);
  if (rc != BBQ_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  std::vector<bbq_index *> shards;
  std::vector<int32_t> devs;
  auto bail = [&](int code) {
    for (bbq_index *s : shards) bbq_index_destroy(s);
    return code;
  };
  for (int s = 0; s < m.h.nShards; ++s) {
    const int32_t dev = n_devices > 0 ? devices[s % n_devices] : s % ndev;  // shard s on device s; fewer devices than shards: round robin
    bbq_index *sh = nullptr;
    rc = bbq_index_load(shard_prefix(prefix, s).c_str(), dev, &sh, nullptr);
    if (rc != BBQ_OK) return bail(rc);
    shards.push_back(sh);
    devs.push_back(dev);
    if (sh->multi || sh->dim != m.h.dimensions || sh->index_bits != m.h.indexBits || sh->row_base != m.bounds[(size_t)2 * s] || sh->n_rows != m.bounds[(size_t)2 * s + 1])
      return bail(fail(BBQ_ERR_INVALID_ARG, "%s: shard %d does not match the manifest", prefix, s));
  }
  rc = multi_assemble(shards.data(), devs.data(), m.h.nShards, m.h.dimensions, m.h.indexBits, m.h.vectorCount, m.h.centroidSquareMagnitude, out);
  if (rc != BBQ_OK) return bail(rc);
  if (centroid_out) memcpy(centroid_out, m.centroid.data(), m.centroid.size() * 4);
  return BBQ_OK;
}

int bbq_index_load(const char *prefix, int32_t device, bbq_index **out, float *centroid_out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load: out is null");
  *out = nullptr;
  if (!prefix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load: null path");
  if (is_manifest(prefix)) return bbq_index_load_multi(prefix, 1, &device, out, centroid_out);  // a multi-device index, every shard on this device
  MetaHeader h;
  PilotExt px{0, 0, 0};
  std::vector<float> cen;
  uint64_t want_sum = 0;
  int rc = read_meta(prefix, &h, &cen, &want_sum, &px);
  if (rc != BBQ_OK) return rc;
  if (h.rowBase + h.vectorCount > 0xFFFFFFFFll) return fail(BBQ_ERR_UNSUPPORTED, "more than 2^32 rows");
  const std::string dpath = std::string(prefix) + ".veb";
  FileCloser fc{fopen(dpath.c_str(), "rb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot open %s", dpath.c_str());
  if (fseeko(fc.f, 0, SEEK_END) != 0 || ftello(fc.f) < h.vectorDataOffset + h.vectorDataLength)
    return fail(BBQ_ERR_INVALID_ARG, "%s: shorter than vectorDataOffset + vectorDataLength", dpath.c_str());
  if (fseeko(fc.f, h.vectorDataOffset, SEEK_SET) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: seek failed", dpath.c_str());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  if (device < 0 || device >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));
  DeviceCtx *ctx = nullptr;
  rc = get_ctx(device, &ctx);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::unique_ptr<bbq_index> ix(new bbq_index());
  ix->device = device;
  ix->dim = h.dimensions;
  ix->index_bits = h.indexBits;
  ix->store_bits = h.dimensions == 1 ? 1 : store_bits_of(h.indexBits);
  ix->pb = row_bytes_of(h.dimensions, ix->store_bits);
  ix->w16 = h.w16;
  ix->n_rows = h.vectorCount;
  ix->row_base = h.rowBase;
  ix->centroid_dp = h.centroidSquareMagnitude;
  ix->has_pilot = px.pilotRows > 0;
  ix->want_compact = h.layout == kLayoutCompact;
  ix->layout = h.layout;
  ix->has_x1 = h.hasX1;
  ix->tile_stride = h.tileStride;
  ix->bytes_per_row = h.tileStride / kTileRows;
  ix->ctx = ctx;
  ix->slots = ctx->slots;
  ix->aux_stream = ctx->aux_stream;
  ix->d_aux_flags = ctx->d_aux_flags;
  rc = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) return rc;
  auto bail = [&](int code) {
    destroy_unlocked(ix.release());
    return code;
  };
  Storage &st = ix->main, &pt = ix->pilot;
  if (h.tilesBytes > 0 && hipMalloc((void **)&st.d_tiles, (size_t)h.tilesBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of tiles", (long long)h.tilesBytes));
  if (h.exactBytes > 0 && hipMalloc((void **)&st.d_exact, (size_t)h.exactBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of exact corrections", (long long)h.exactBytes));
  if (px.pilotTilesBytes > 0 && hipMalloc((void **)&pt.d_tiles, (size_t)px.pilotTilesBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of pilot tiles", (long long)px.pilotTilesBytes));
  if (px.pilotExactBytes > 0 && hipMalloc((void **)&pt.d_exact, (size_t)px.pilotExactBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of pilot corrections", (long long)px.pilotExactBytes));
  const size_t piece = 64u << 20;
  std::vector<uint8_t> buf(piece);
  uint8_t *dst[4] = {st.d_tiles, (uint8_t *)st.d_exact, pt.d_tiles, (uint8_t *)pt.d_exact};
  const int64_t len[4] = {h.tilesBytes, h.exactBytes, px.pilotTilesBytes, px.pilotExactBytes};
  uint64_t dsum = kFnvSeed;
  for (int part = 0; part < 4; ++part) {
    for (int64_t o = 0; o < len[part]; o += (int64_t)piece) {
      const size_t m = (size_t)std::min<int64_t>((int64_t)piece, len[part] - o);
      if (fread(buf.data(), 1, m, fc.f) != m) return bail(fail(BBQ_ERR_INVALID_ARG, "%s: read failed", dpath.c_str()));
      dsum = fnv64_words(buf.data(), m, dsum);
      if (hipMemcpy(dst[part] + o, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(BBQ_ERR_HIP, "bbq_index_load: copy to the device failed"));
    }
  }
  if (dsum != want_sum) return bail(fail(BBQ_ERR_INVALID_ARG, "%s: vector data checksum mismatch", dpath.c_str()));
  auto fill_view = [&](Storage &sto, int64_t rows, int64_t row_id_base) {
    sto.row_id_base = row_id_base;
    sto.view.n_rows = rows;
    sto.view.w16 = h.w16;
    sto.view.tile_stride = h.tileStride;
    sto.view.has_x1 = h.hasX1;
    sto.view.dim = h.dimensions;
    sto.view.layout = h.layout;
    sto.view.store_bits = ix->store_bits;
    sto.view.tiles = sto.d_tiles;
    sto.view.exact = sto.d_exact;
    sto.view.add_range = add_range_of(sto.d_exact, (rows + kTileRows - 1) / kTileRows);
  };
  fill_view(st, h.vectorCount, h.rowBase);
  if (ix->has_pilot) fill_view(pt, px.pilotRows, 0);
  if (centroid_out) memcpy(centroid_out, cen.data(), cen.size() * 4);
  *out = ix.release();
  return BBQ_OK;
}

int bbq_index_export(bbq_index *ix, uint8_t *codes, double *corr) {
  clear_error();
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_export: null handle");
  if (ix->n_rows == 0 || (!codes && !corr)) return BBQ_OK;
  if (ix->multi) return multi_export(ix, codes, corr);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  const int64_t n_tiles = (ix->n_rows + kTileRows - 1) / kTileRows;
  const int64_t group = std::max<int64_t>(1, (64ll << 20) / ix->tile_stride);  // tiles per piece
  std::vector<uint8_t> buf((size_t)(group * ix->tile_stride));
  std::vector<double> ex;
  if (corr && ix->layout == kLayoutCompact) ex.resize((size_t)(group * kTileRows * 4));
  const int pb = ix->pb, w16 = ix->w16;
  for (int64_t t0 = 0; t0 < n_tiles; t0 += group) {
    const int64_t nt = std::min(group, n_tiles - t0);
    HIPCHK(hipMemcpy(buf.data(), ix->main.d_tiles + t0 * ix->tile_stride, (size_t)(nt * ix->tile_stride), hipMemcpyDeviceToHost));
    if (!ex.empty())
      HIPCHK(hipMemcpy(ex.data(), ix->main.d_exact + t0 * kTileRows * 4, (size_t)(nt * kTileRows) * 32, hipMemcpyDeviceToHost));
    for (int64_t t = 0; t < nt; ++t) {
      const uint8_t *tp = buf.data() + t * ix->tile_stride;
      const uint8_t *cr = tp + (size_t)w16 * (kTileRows * 16);
      for (int r = 0; r < kTileRows; ++r) {
        const int64_t row = (t0 + t) * kTileRows + r;
        if (row >= ix->n_rows) break;
        int ones = 0;  // popcount of a 1-bit row / component sum of a multi-bit row
        if (ix->store_bits == 1) {
          for (int b = 0; b < pb; ++b) {
            const uint8_t v = tp[((size_t)(b >> 4) * kTileRows + r) * 16 + (b & 15)];
            if (codes) codes[row * pb + b] = v;
            ones += __builtin_popcount(v);
          }
        } else {  // fields back to one byte per dimension (src/binaryQuantizationFormat.ts:241-245)
          const int sb = ix->store_bits;
          for (int d = 0; d < ix->dim; ++d) {
            const int bit = d * sb, b = bit >> 3;
            const uint8_t byte = tp[((size_t)(b >> 4) * kTileRows + r) * 16 + (b & 15)];
            const uint8_t v = (uint8_t)((byte >> (bit & 7)) & ((1u << sb) - 1u));
            if (codes) codes[row * (int64_t)ix->dim + d] = v;
            ones += v;
          }
        }
        if (!corr) continue;
        double *c = corr + row * 4;
        if (ix->layout == kLayoutCompact) {
          const double *e = ex.data() + ((size_t)t * kTileRows + r) * 4;
          c[0] = e[0]; c[1] = e[1]; c[2] = e[2];
          c[3] = (double)ones;  // compact layout is only chosen when every sum equals the popcount
        } else {
          memcpy(c, cr + r * 16, 16);
          memcpy(c + 2, cr + 1024 + r * 8, 8);
          if (ix->has_x1) memcpy(c + 3, cr + 1536 + r * 8, 8);
          else c[3] = (double)ones;
        }
      }
    }
  }
  return BBQ_OK;
}

}  // extern "C"
