// gob.hpp -- the part of Go's encoding/gob wire format that the reference's model files use, so that files written
// here can be read by the reference and vice versa: encoding.WriteGob / ReadGob (common/encoding/encoding.go:98-120:
// a little-endian int32 byte count, then one gob stream) of
//   * model.Params = map[ParamName]any           (BaseMatrixFactorization.Marshal, model/cf/model.go:206-211)
//   * int64 / int / string                       (logics/cf.go:86-107, 148-152)
// encoding/gob is the Go standard library (not under /root/reference, no Go toolchain in this image): restated from the
// package documentation's wire-format grammar -- PARITY UNPINNED against a real Go encoder.  What is checked
// (tests/test_host_mirror_cpu.py): the documentation's worked example (a struct Point{22, 33}: 1f ff 81 03 ... / 07 ff 82
// 01 2c 01 42 00) rebuilt from the primitives below, the integer / float rules on known values, and round trips.
//
// Wire rules used (package doc, "Encoding Details"):
//   uint    < 128: one byte; else one byte holding the NEGATED byte count, then the big-endian bytes, minimal length
//   int     zig-zag-like: bit 0 = complement flag, the rest the (possibly complemented) value; sent as a uint
//   float   the IEEE-754 float64 bits BYTE-REVERSED, sent as a uint (float32 is widened first)
//   string  uint length + bytes;   bool = uint 0 / 1
//   message uint byte count, then int type id (negative = a type definition for -id follows), then the data
//   a top-level or interface-held value that is not a struct is preceded by one zero byte
//   interface value: string name of the concrete type ("" = nil), int type id of it, uint byte count, then the value
//   predefined type ids: bool 1, int 2, uint 3, float 4, []byte 5, string 6, complex 7, interface 8; user types from 65
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gorse {
namespace gob {

enum TypeId { tBool = 1, tInt = 2, tUint = 3, tFloat = 4, tBytes = 5, tString = 6, tComplex = 7, tInterface = 8, tFirstUser = 65 };

inline void put_uint(std::string &b, uint64_t v) {
    if (v < 128) {
        b.push_back((char)v);
        return;
    }
    int n = 8;
    while (n > 1 && ((v >> (8 * (n - 1))) & 0xFF) == 0) n--;
    b.push_back((char)(uint8_t)(-n));
    for (int k = n - 1; k >= 0; k--) b.push_back((char)((v >> (8 * k)) & 0xFF));
}
inline void put_int(std::string &b, int64_t i) {
    const uint64_t u = i < 0 ? ((~(uint64_t)i) << 1) | 1 : ((uint64_t)i << 1);
    put_uint(b, u);
}
inline void put_float(std::string &b, double f) {
    uint64_t bits, rev = 0;
    std::memcpy(&bits, &f, 8);
    for (int k = 0; k < 8; k++) rev = (rev << 8) | ((bits >> (8 * k)) & 0xFF);
    put_uint(b, rev);
}
inline void put_string(std::string &b, const std::string &s) {
    put_uint(b, s.size());
    b += s;
}
// one length-prefixed message
inline std::string message(const std::string &body) {
    std::string out;
    put_uint(out, body.size());
    return out + body;
}

struct Reader {
    const uint8_t *p, *end;
    Reader(const std::string &s) : p((const uint8_t *)s.data()), end((const uint8_t *)s.data() + s.size()) {}
    Reader(const uint8_t *a, const uint8_t *b) : p(a), end(b) {}
    bool done() const { return p >= end; }
    uint8_t byte() {
        if (p >= end) throw std::runtime_error("gob: unexpected end of data");
        return *p++;
    }
    uint64_t uint() {
        const uint8_t b0 = byte();
        if (b0 < 128) return b0;
        const int n = -(int)(int8_t)b0;
        if (n < 1 || n > 8) throw std::runtime_error("gob: bad unsigned integer length");
        uint64_t v = 0;
        for (int k = 0; k < n; k++) v = (v << 8) | byte();
        return v;
    }
    int64_t sint() {
        const uint64_t u = uint();
        return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
    }
    double flt() {
        const uint64_t rev = uint();
        uint64_t bits = 0;
        for (int k = 0; k < 8; k++) bits = (bits << 8) | ((rev >> (8 * k)) & 0xFF);
        double f;
        std::memcpy(&f, &bits, 8);
        return f;
    }
    std::string str() {
        const uint64_t n = uint();
        if ((uint64_t)(end - p) < n) throw std::runtime_error("gob: string runs past the end");
        std::string s((const char *)p, (size_t)n);
        p += n;
        return s;
    }
    Reader sub(uint64_t n) {
        if ((uint64_t)(end - p) < n) throw std::runtime_error("gob: message runs past the end");
        Reader r(p, p + n);
        p += n;
        return r;
    }
};

// ---- top-level singletons: gob.NewEncoder(buf).Encode(v) for v int / int64 / string --------------------------------
inline std::string encode_int(int64_t v) {
    std::string body;
    put_int(body, tInt);
    body.push_back(0);
    put_int(body, v);
    return message(body);
}
inline std::string encode_string(const std::string &s) {
    std::string body;
    put_int(body, tString);
    body.push_back(0);
    put_string(body, s);
    return message(body);
}
// the value message of a stream, type definitions skipped; returns its reader positioned after the type id
inline Reader value_message(Reader &r, int64_t &type_id) {
    for (;;) {
        Reader m = r.sub(r.uint());
        type_id = m.sint();
        if (type_id >= 0) return m;  // negative: the definition of type -id, nothing we need to keep
    }
}
inline int64_t decode_int(const std::string &stream) {
    Reader r(stream);
    int64_t id;
    Reader m = value_message(r, id);
    if (id != tInt && id != tUint) throw std::runtime_error("gob: expected an integer, got type id " + std::to_string(id));
    if (m.byte() != 0) throw std::runtime_error("gob: singleton marker missing");
    return id == tInt ? m.sint() : (int64_t)m.uint();
}
inline std::string decode_string(const std::string &stream) {
    Reader r(stream);
    int64_t id;
    Reader m = value_message(r, id);
    if (id != tString) throw std::runtime_error("gob: expected a string, got type id " + std::to_string(id));
    if (m.byte() != 0) throw std::runtime_error("gob: singleton marker missing");
    return m.str();
}

// ---- []float32 at top level (common/ann/hnsw.go:299-303: encoding.WriteGob of every stored vector) -----------------------
// Two messages: the definition of type 65 as wireType{SliceT: sliceType{CommonType{Name: "[]float32", Id: 65}, Elem: float}},
// then the value: type id 65, the singleton marker 0, the element count, the elements as gob floats.
inline std::string encode_f32_slice(const float *v, size_t n) {
    std::string def;
    put_int(def, -65);
    put_uint(def, 2);  // wireType field 1 (SliceT): delta 2 from -1
    put_uint(def, 1);  //   sliceType field 0 (CommonType)
    put_uint(def, 1);  //     CommonType field 0 (Name)
    put_string(def, "[]float32");
    put_uint(def, 1);  //     CommonType field 1 (Id)
    put_int(def, 65);
    put_uint(def, 0);  //     end of CommonType
    put_uint(def, 1);  //   sliceType field 1 (Elem)
    put_int(def, tFloat);
    put_uint(def, 0);  //   end of sliceType
    put_uint(def, 0);  // end of wireType
    std::string val;
    put_int(val, 65);
    put_uint(val, 0);  // not a struct: singleton marker
    put_uint(val, n);
    for (size_t k = 0; k < n; k++) put_float(val, (double)v[k]);
    return message(def) + message(val);
}
inline std::vector<float> decode_f32_slice(const std::string &stream) {
    Reader r(stream);
    int64_t id;
    Reader m = value_message(r, id);
    if (m.uint() != 0) throw std::runtime_error("gob: expected a non-struct value");
    const uint64_t n = m.uint();
    if (n > (uint64_t)(m.end - m.p)) throw std::runtime_error("gob: slice length runs past the end");  // >= 1 byte per element
    std::vector<float> out((size_t)n);
    for (auto &x : out) x = (float)m.flt();
    return out;
}
// time.Time at top level travels as a GobEncoder value: the last bytes of the stream are time.Time.MarshalBinary -- version
// 1: 1 + 8 (seconds since year 1, big endian) + 4 (nanoseconds) + 2 (zone offset in minutes, -1 = UTC) = 15 bytes; version 2
// adds one byte of offset seconds.  Returns Unix nanoseconds; the zone is dropped (an instant, not a wall clock).
inline int64_t decode_time_unix_nanos(const std::string &stream) {
    auto parse = [&](size_t len, uint8_t version, int64_t &out) {
        if (stream.size() < len + 1) return false;
        const uint8_t *b = (const uint8_t *)stream.data() + stream.size() - len;
        if (b[0] != version || b[-1] != len) return false;  // preceded by its byte count
        uint64_t sec = 0, ns = 0;
        for (int k = 0; k < 8; k++) sec = (sec << 8) | b[1 + k];
        for (int k = 0; k < 4; k++) ns = (ns << 8) | b[9 + k];
        const int64_t unix_sec = (int64_t)sec - 62135596800ll;  // seconds from year 1 to 1970
        out = unix_sec * 1000000000ll + (int64_t)ns;
        return true;
    };
    int64_t t = 0;
    if (parse(15, 1, t) || parse(16, 2, t)) return t;
    throw std::runtime_error("gob: not a time.Time value");
}

// gob.NewEncoder(buf).Encode(t) for a time.Time in UTC: the definition of type 65 as wireType{GobEncoderT (field 4):
// gobEncoderType{CommonType{Name: "Time", Id: 65}}}, then type id 65, the singleton marker, the byte count and
// time.Time.MarshalBinary version 1 (zone offset -1 = UTC).
inline std::string encode_time_unix_nanos(int64_t t) {
    std::string def;
    put_int(def, -65);
    put_uint(def, 5);  // wireType field 4 (GobEncoderT): delta 5 from -1
    put_uint(def, 1);  //   gobEncoderType field 0 (CommonType)
    put_uint(def, 1);  //     CommonType field 0 (Name)
    put_string(def, "Time");
    put_uint(def, 1);  //     CommonType field 1 (Id)
    put_int(def, 65);
    put_uint(def, 0);  //     end of CommonType
    put_uint(def, 0);  //   end of gobEncoderType
    put_uint(def, 0);  // end of wireType
    int64_t sec = t / 1000000000ll, ns = t % 1000000000ll;
    if (ns < 0) ns += 1000000000ll, sec -= 1;
    const uint64_t s1 = (uint64_t)(sec + 62135596800ll);
    std::string payload(1, (char)1);
    for (int k = 7; k >= 0; k--) payload.push_back((char)((s1 >> (8 * k)) & 0xFF));
    for (int k = 3; k >= 0; k--) payload.push_back((char)(((uint64_t)ns >> (8 * k)) & 0xFF));
    payload += "\xff\xff";
    std::string val;
    put_int(val, 65);
    put_uint(val, 0);
    put_string(val, payload);
    return message(def) + message(val);
}

// ---- model.Params = map[ParamName]any ------------------------------------------------------------------------------
struct Value {
    enum Kind { Int, Float, Bool, String } kind = Float;
    int64_t i = 0;
    double f = 0;
    bool b = false;
    std::string s;
    static Value of_int(int64_t v) {
        Value x;
        x.kind = Int, x.i = v, x.f = (double)v;
        return x;
    }
    static Value of_float(double v) {
        Value x;
        x.kind = Float, x.f = v;
        return x;
    }
    double number() const { return kind == Int ? (double)i : kind == Bool ? (b ? 1.0 : 0.0) : f; }
};
using Entries = std::vector<std::pair<std::string, Value>>;

// Two messages, as gob sends a value of a type it has not described yet: the definition of type 65 as
// wireType{MapT: mapType{CommonType{Name: type_name, Id: 65}, Key: string, Elem: interface}}, then the map.
inline std::string encode_map(const std::string &type_name, const Entries &entries) {
    std::string def;
    put_int(def, -(int64_t)tFirstUser);
    put_uint(def, 4);  // wireType field 3 (MapT): delta 4 from -1
    put_uint(def, 1);  //   mapType field 0 (CommonType)
    put_uint(def, 1);  //     CommonType field 0 (Name)
    put_string(def, type_name);
    put_uint(def, 1);  //     CommonType field 1 (Id)
    put_int(def, tFirstUser);
    put_uint(def, 0);  //     end of CommonType
    put_uint(def, 1);  //   mapType field 1 (Key)
    put_int(def, tString);
    put_uint(def, 1);  //   mapType field 2 (Elem)
    put_int(def, tInterface);
    put_uint(def, 0);  //   end of mapType
    put_uint(def, 0);  // end of wireType
    std::string val;
    put_int(val, tFirstUser);
    put_uint(val, 0);  // not a struct: singleton marker
    put_uint(val, entries.size());
    for (const auto &kv : entries) {
        put_string(val, kv.first);
        std::string inner;
        inner.push_back(0);  // the concrete value is not a struct either
        const Value &v = kv.second;
        switch (v.kind) {
            case Value::Int:
                put_string(val, "int");
                put_int(val, tInt);
                put_int(inner, v.i);
                break;
            case Value::Float:
                put_string(val, "float64");
                put_int(val, tFloat);
                put_float(inner, v.f);
                break;
            case Value::Bool:
                put_string(val, "bool");
                put_int(val, tBool);
                put_uint(inner, v.b ? 1 : 0);
                break;
            case Value::String:
                put_string(val, "string");
                put_int(val, tString);
                put_string(inner, v.s);
                break;
        }
        put_uint(val, inner.size());
        val += inner;
    }
    return message(def) + message(val);
}

inline Entries decode_map(const std::string &stream) {
    Reader r(stream);
    int64_t id;
    Reader m = value_message(r, id);
    if (id < tFirstUser) throw std::runtime_error("gob: expected a map, got type id " + std::to_string(id));
    if (m.byte() != 0) throw std::runtime_error("gob: singleton marker missing");
    const uint64_t n = m.uint();
    Entries out;
    for (uint64_t k = 0; k < n; k++) {
        std::string key = m.str();
        const std::string name = m.str();
        Value v;
        if (name.empty()) {  // nil interface: keep the key with a zero value
            out.emplace_back(std::move(key), Value::of_float(0));
            continue;
        }
        const int64_t cid = m.sint();
        if (cid < 0 || cid >= tFirstUser) throw std::runtime_error("gob: value of " + key + " has the unsupported type " + name);
        Reader in = m.sub(m.uint());
        if (in.byte() != 0) throw std::runtime_error("gob: singleton marker missing in " + key);
        if (cid == tInt) {
            v = Value::of_int(in.sint());
        } else if (cid == tUint) {
            v = Value::of_int((int64_t)in.uint());
        } else if (cid == tFloat) {
            v = Value::of_float(in.flt());
        } else if (cid == tBool) {
            v.kind = Value::Bool, v.b = in.uint() != 0;
        } else if (cid == tString) {
            v.kind = Value::String, v.s = in.str();
        } else {
            throw std::runtime_error("gob: value of " + key + " has the unsupported type " + name);
        }
        out.emplace_back(std::move(key), std::move(v));
    }
    return out;
}

}  // namespace gob
}  // namespace gorse
