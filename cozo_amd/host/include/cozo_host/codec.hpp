// codec.hpp -- the byte forms of stored rows (C++ host mirror): what the storage iterator hands to cozo-core and what
// `store_tx.put` takes.
//   key   = 8-byte big-endian relation id + key columns in the memcmp encoding
//           (MemCmpEncoder::encode_datavalue / DataValue::decode_from_key, data/memcmp.rs:46-163, 258-365;
//            RelationHandle::encode_key_for_store, runtime/relation.rs:247-267; decode_tuple_from_key, data/tuple.rs:41-52)
//   value = 8-byte prefix + ONE msgpack array of the non-key columns in rmp-serde 1.2.0's representation of
//           `enum DataValue` (encode_val_for_store, runtime/relation.rs:275-296; extend_tuple_from_v, :526-531).
//           rmp-serde is a crates.io dependency (Cargo.lock:3224-3226), not vendored: restated from its published
//           behaviour, "parity unpinned" (the reference's tests hold no stored bytes).
// Only the variants value.hpp models (Null, Bool, Num, Str, Bytes, List, Vec F32) are encoded / decoded.
// StoredRows is a relation as a scan yields it; FixedRuleInputRelation::from_stored hands it to libcozo_ingest
// (include/cozo_ingest.h) instead of decoding every row.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "cozo_ingest.h"
#include "value.hpp"

namespace cozo {

struct CodecError : std::runtime_error {
    explicit CodecError(const std::string &msg) : std::runtime_error(msg) {}
};

// ---- memcmp ---------------------------------------------------------------------------------------------------
void encode_datavalue(std::vector<uint8_t> &out, const DataValue &v);
void encode_bytes(std::vector<uint8_t> &out, const uint8_t *key, size_t len);                 // memcmp.rs:147-163
std::vector<uint8_t> decode_bytes(const uint8_t *&p, const uint8_t *end);                     // memcmp.rs:165-192
DataValue decode_datavalue(const uint8_t *&p, const uint8_t *end);                            // advances p
inline std::vector<uint8_t> memcmp_bytes(const DataValue &v) {
    std::vector<uint8_t> out;
    encode_datavalue(out, v);
    return out;
}

// ---- stored rows ----------------------------------------------------------------------------------------------
std::vector<uint8_t> encode_key_for_store(uint64_t relation_id, const Tuple &t, size_t n_key_cols);
std::vector<uint8_t> encode_val_for_store(uint64_t relation_id, const Tuple &t, size_t n_key_cols);
Tuple decode_tuple_from_key(const std::vector<uint8_t> &key);
Tuple decode_tuple_from_kv(const uint8_t *key, size_t key_len, const uint8_t *val, size_t val_len);

struct StoredRows {
    std::vector<uint8_t> keys, vals;
    std::vector<uint64_t> key_off{0}, val_off{0};
    uint32_t n_key_cols = 0;

    size_t size() const { return key_off.size() - 1; }
    // the write path: what a sequence of store_tx.put calls leaves behind (rows ordered by key bytes, a later put of a
    // key replaces the earlier one)
    static StoredRows from_tuples(uint64_t relation_id, const std::vector<Tuple> &tuples, uint32_t n_key_cols);
    Tuple tuple(size_t i) const {
        return decode_tuple_from_kv(keys.data() + key_off[i], key_off[i + 1] - key_off[i], vals.data() + val_off[i],
                                    val_off[i + 1] - val_off[i]);
    }
    czi_rows view() const { return czi_rows{keys.data(), key_off.data(), vals.data(), val_off.data(), size(), n_key_cols}; }
};

// What a statement has to write to turn the stored rows `old_rows` into `new_rows` (both ascending by key bytes, as a scan
// yields them): `puts` = the rows of new_rows whose key is new or whose value bytes differ, `dels` = the keys only old_rows
// has.  The write-back of index maintenance on the device: the encoded `tbl:idx` rows after cz_hnsw_insert / cz_hnsw_remove
// against the rows the store holds.
void stored_rows_delta(const StoredRows &old_rows, const StoredRows &new_rows, StoredRows *puts, std::vector<std::vector<uint8_t>> *dels);

}  // namespace cozo
