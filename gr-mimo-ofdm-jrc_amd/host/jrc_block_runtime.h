// jrc_block_runtime.h — the block-runtime surface the host-side blocks (jrc_blocks.h/.cc) are written against.
//
// With -DJRC_WITH_GNURADIO this header just pulls in GNU Radio 3.8 (gr::block, gr::tagged_stream_block, pmt) and the
// blocks are real GNU Radio blocks.  Without it (this image has no GNU Radio) it provides a small stand-alone runtime
// of our own with the same member names, so that the very same work()/general_work() bodies can be driven by the test
// harness (jrc_blocks_capi.cc -> tests/test_host_blocks.py).  It is product test scaffolding for OUR blocks; it is not
// used to build anything from the reference.
#pragma once

#ifdef JRC_WITH_GNURADIO

#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/tagged_stream_block.h>
#include <pmt/pmt.h>
namespace jrc_rt = gr;
#define JRC_SPTR boost::shared_ptr
#define JRC_GET_INITIAL_SPTR(p) gnuradio::get_initial_sptr(p)

#else  // ---------------------------------------------------------------- stand-alone runtime

#include <complex>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <deque>
#include <map>
#include <utility>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

namespace pmt {
struct node;
typedef std::shared_ptr<node> pmt_t;
struct node {
    enum kind_t { NIL, SYMBOL, LONG, U64, DOUBLE, F32VEC, C32VEC, LIST, DICT, BLOB, PAIR } kind = NIL;
    std::string s; long l = 0; uint64_t u = 0; double d = 0;
    std::vector<float> f; std::vector<gr_complex> c; std::vector<uint8_t> blob;
    std::vector<pmt_t> list; std::vector<std::pair<std::string, pmt_t>> dict;
};
inline pmt_t mk(node::kind_t k) { auto p = std::make_shared<node>(); p->kind = k; return p; }
inline pmt_t string_to_symbol(const std::string& s) { auto p = mk(node::SYMBOL); p->s = s; return p; }
inline pmt_t mp(const std::string& s) { return string_to_symbol(s); }
inline pmt_t intern(const std::string& s) { return string_to_symbol(s); }
inline std::string symbol_to_string(const pmt_t& p) { return p->s; }
inline pmt_t from_long(long v) { auto p = mk(node::LONG); p->l = v; return p; }
inline pmt_t from_uint64(uint64_t v) { auto p = mk(node::U64); p->u = v; return p; }
inline pmt_t from_double(double v) { auto p = mk(node::DOUBLE); p->d = v; return p; }
inline long to_long(const pmt_t& p) { return p->kind == node::U64 ? (long)p->u : (p->kind == node::DOUBLE ? (long)p->d : p->l); }
inline uint64_t to_uint64(const pmt_t& p) { return p->kind == node::LONG ? (uint64_t)p->l : p->u; }
inline double to_double(const pmt_t& p) { return p->kind == node::LONG ? (double)p->l : (p->kind == node::U64 ? (double)p->u : p->d); }
inline float to_float(const pmt_t& p) { return (float)to_double(p); }
inline pmt_t init_f32vector(size_t n, const float* v) { auto p = mk(node::F32VEC); p->f.assign(v, v + n); return p; }
inline pmt_t init_c32vector(size_t n, const gr_complex* v) { auto p = mk(node::C32VEC); p->c.assign(v, v + n); return p; }
inline pmt_t list2(pmt_t a, pmt_t b) { auto p = mk(node::LIST); p->list = {a, b}; return p; }
inline pmt_t list4(pmt_t a, pmt_t b, pmt_t c, pmt_t d) { auto p = mk(node::LIST); p->list = {a, b, c, d}; return p; }
inline pmt_t make_tuple(pmt_t a, pmt_t b) { return list2(a, b); }
inline pmt_t make_dict() { return mk(node::DICT); }
inline pmt_t make_blob(const void* p, size_t n) { auto b = mk(node::BLOB); b->blob.assign((const uint8_t*)p, (const uint8_t*)p + n); return b; }
inline pmt_t cons(pmt_t a, pmt_t b) { auto p = mk(node::PAIR); p->list = {a, b}; return p; }
inline pmt_t car(const pmt_t& p) { return p->list[0]; }
inline pmt_t cdr(const pmt_t& p) { return p->list[1]; }
inline bool is_symbol(const pmt_t& p) { return p && p->kind == node::SYMBOL; }
inline bool is_pair(const pmt_t& p) { return p && p->kind == node::PAIR; }
inline size_t blob_length(const pmt_t& p) { return p->blob.size(); }
inline const void* blob_data(const pmt_t& p) { return p->blob.data(); }
inline float to_float(const pmt_t& p);
inline std::vector<gr_complex> c32vector_elements(const pmt_t& p) { return p && p->kind == node::C32VEC ? p->c : std::vector<gr_complex>(); }
inline pmt_t dict_ref(const pmt_t& dct, const pmt_t& key, const pmt_t& not_found)
{
    if (dct && dct->kind == node::DICT)
        for (auto& kv : dct->dict) if (kv.first == key->s) return kv.second;
    return not_found;
}
inline pmt_t dict_add(const pmt_t& dct, const pmt_t& key, const pmt_t& val)
{
    auto p = std::make_shared<node>(*dct);
    p->dict.emplace_back(key->s, val);
    return p;
}
inline void to_json(const pmt_t& p, std::ostringstream& o)
{
    o.precision(17);
    if (!p) { o << "null"; return; }
    switch (p->kind) {
        case node::NIL: o << "null"; break;
        case node::SYMBOL: o << '"' << p->s << '"'; break;
        case node::LONG: o << p->l; break;
        case node::U64: o << p->u; break;
        case node::DOUBLE:
            if (p->d != p->d) o << "\"nan\""; else if (p->d > 1e308) o << "\"inf\""; else if (p->d < -1e308) o << "\"-inf\""; else o << p->d;
            break;
        case node::F32VEC: o << '['; for (size_t i = 0; i < p->f.size(); i++) { if (i) o << ','; o << (double)p->f[i]; } o << ']'; break;
        case node::C32VEC: o << '['; for (size_t i = 0; i < p->c.size(); i++) { if (i) o << ','; o << '[' << (double)p->c[i].real() << ',' << (double)p->c[i].imag() << ']'; } o << ']'; break;
        case node::LIST: o << '['; for (size_t i = 0; i < p->list.size(); i++) { if (i) o << ','; to_json(p->list[i], o); } o << ']'; break;
        case node::BLOB: o << "{\"blob\":["; for (size_t i = 0; i < p->blob.size(); i++) { if (i) o << ','; o << (int)p->blob[i]; } o << "]}"; break;
        case node::PAIR: o << "{\"car\":"; to_json(p->list[0], o); o << ",\"cdr\":"; to_json(p->list[1], o); o << '}'; break;
        case node::DICT: o << '{'; for (size_t i = 0; i < p->dict.size(); i++) { if (i) o << ','; o << '"' << p->dict[i].first << "\":"; to_json(p->dict[i].second, o); } o << '}'; break;
    }
}
}  // namespace pmt

namespace jrc_host {

struct tag_t { uint64_t offset = 0; pmt::pmt_t key, value, srcid; };

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    int min_streams, max_streams; std::vector<int> sizes;
    static sptr make(int mn, int mx, int size) { auto p = std::make_shared<io_signature>(); p->min_streams = mn; p->max_streams = mx; p->sizes = {size}; return p; }
    static sptr make3(int mn, int mx, int s0, int s1, int s2) { auto p = std::make_shared<io_signature>(); p->min_streams = mn; p->max_streams = mx; p->sizes = {s0, s1, s2}; return p; }
};

namespace thread { typedef std::mutex mutex; typedef std::unique_lock<std::mutex> scoped_lock; }

class block {
public:
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2 };
    block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out)
    {
        const int ni = in->max_streams > 0 ? in->max_streams : 0, no = out->max_streams > 0 ? out->max_streams : 0;
        t_in_tags.resize(ni); t_read.assign(ni, 0); t_consumed.assign(ni, 0);
        t_out_tags.resize(no); t_written.assign(no, 0);
    }
    block() : d_name("?"), d_in(io_signature::make(0, 0, 0)), d_out(io_signature::make(0, 0, 0)) {}   // for virtual bases
    virtual ~block() {}
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;
    virtual void forecast(int noutput_items, gr_vector_int& req) { for (auto& r : req) r = noutput_items; }
    virtual bool start() { return true; }                         // gr::block::start / stop: called by the scheduler around a run
    virtual bool stop() { return true; }
    std::string name() const { return d_name; }
    std::string alias() const { return d_name + "0"; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }

    // ---- test hooks (what the scheduler would own) ----
    std::vector<std::vector<tag_t>> t_in_tags, t_out_tags;
    std::vector<uint64_t> t_read, t_written;
    std::vector<int> t_consumed;
    std::vector<std::pair<std::string, pmt::pmt_t>> t_published;
    std::mutex t_pub_lock;                                        // guards t_published
    std::map<std::string, std::deque<pmt::pmt_t>> t_msg_in;     // message input queues (what the scheduler delivers)
    // one scheduler turn: call general_work, then advance the item counters like the scheduler does
    virtual int t_run(int noutput_items, gr_vector_int& nin, gr_vector_const_void_star& in, gr_vector_void_star& out)
    {
        std::fill(t_consumed.begin(), t_consumed.end(), 0);
        int n = general_work(noutput_items, nin, in, out);
        for (size_t i = 0; i < t_read.size(); i++) t_read[i] += t_consumed[i];
        if (n > 0) for (auto& w : t_written) w += n;
        for (size_t i = 0; i < t_in_tags.size() && i < t_read.size(); i++) {      // the scheduler drops the tags of consumed items
            auto& v = t_in_tags[i];
            v.erase(std::remove_if(v.begin(), v.end(), [&](const tag_t& t) { return t.offset < t_read[i]; }), v.end());
        }
        return n;
    }

protected:
    void set_tag_propagation_policy(tag_propagation_policy_t) {}
    void set_relative_rate(double) {}
    void set_relative_rate(uint64_t, uint64_t) {}
    static int set_thread_priority(int) { return 0; }
    uint64_t nitems_read(unsigned i) const { return t_read[i]; }
    uint64_t nitems_written(unsigned i) const { return t_written[i]; }
    void add_item_tag(unsigned port, uint64_t off, const pmt::pmt_t& key, const pmt::pmt_t& val,
                      const pmt::pmt_t& srcid = pmt::pmt_t())
    {
        tag_t t; t.offset = off; t.key = key; t.value = val; t.srcid = srcid;
        t_out_tags[port].push_back(t);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned port, uint64_t start, uint64_t end)
    {
        v.clear();
        for (auto& t : t_in_tags[port]) if (t.offset >= start && t.offset < end) v.push_back(t);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned port, uint64_t start, uint64_t end, const pmt::pmt_t& key)
    {
        v.clear();
        for (auto& t : t_in_tags[port]) if (t.offset >= start && t.offset < end && t.key->s == key->s) v.push_back(t);
    }
    void get_tags_in_window(std::vector<tag_t>& v, unsigned port, uint64_t rs, uint64_t re, const pmt::pmt_t& key)
    {
        get_tags_in_range(v, port, nitems_read(port) + rs, nitems_read(port) + re, key);
    }
    void consume(int port, int n) { t_consumed[port] += n; }
    void consume_each(int n) { for (auto& c : t_consumed) c += n; }
    void message_port_register_out(const pmt::pmt_t&) {}
    void message_port_register_in(const pmt::pmt_t&) {}
    pmt::pmt_t delete_head_nowait(const pmt::pmt_t& port)
    {
        auto& q = t_msg_in[port->s];
        if (q.empty()) return pmt::pmt_t();
        auto m = q.front(); q.pop_front();
        return m;
    }
    void message_port_pub(const pmt::pmt_t& port, const pmt::pmt_t& msg)      // may be called from a block's own thread, like gr::basic_block's
    {
        std::lock_guard<std::mutex> g(t_pub_lock);
        t_published.emplace_back(port->s, msg);
    }
    double pc_output_buffers_full(int) { return 0; }
    thread::mutex d_setlock;

private:
    std::string d_name;
    io_signature::sptr d_in, d_out;
};

// GNU Radio 3.8 tagged_stream_block contract: the packet length comes from the len_key tag on the
// first item, work() sees exactly one packet, packets are consumed by the base class, which also writes the output
// length tag.
class tagged_stream_block : public block {
public:
    tagged_stream_block(const std::string& name, io_signature::sptr in, io_signature::sptr out, const std::string& len_key)
        : block(name, in, out), d_length_tag_key_str(len_key) {}
    tagged_stream_block() {}
    virtual int work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) = 0;
    virtual int calculate_output_stream_length(const gr_vector_int& ninput_items) = 0;
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        gr_vector_int n_in(ninput_items.size(), 0);
        for (size_t i = 0; i < ninput_items.size(); i++) {
            std::vector<tag_t> tags;
            get_tags_in_range(tags, (unsigned)i, nitems_read((unsigned)i), nitems_read((unsigned)i) + 1, pmt::mp(d_length_tag_key_str));
            if (tags.empty()) return 0;                       // no length tag yet: wait
            n_in[i] = (int)pmt::to_long(tags[0].value);
            if (n_in[i] > ninput_items[i]) return 0;          // packet not complete yet
        }
        if (calculate_output_stream_length(n_in) > noutput_items) return 0;
        int n = work(noutput_items, n_in, input_items, output_items);
        for (size_t i = 0; i < n_in.size(); i++) consume((int)i, n_in[i]);
        if (n > 0)
            for (size_t o = 0; o < t_out_tags.size(); o++)
                add_item_tag((unsigned)o, nitems_written((unsigned)o), pmt::mp(d_length_tag_key_str), pmt::from_long(n));
        return n;
    }

protected:
    void update_length_tags(int, int) {}                      // no-op inside work()
    std::string d_length_tag_key_str;
};

// GNU Radio sync_block contract: one output item per input item; with set_history(h) the input pointer starts h-1 items
// before the first new item and the scheduler only calls work() when noutput_items + h - 1 input items are there.
class sync_block : public block {
public:
    sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : block(name, in, out) {}
    sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        int n = noutput_items;
        for (int ni : ninput_items) n = std::min(n, ni - (int)(d_history - 1));
        if (n <= 0) return 0;
        int r = work(n, input_items, output_items);
        if (r > 0) consume_each(r);
        return r;
    }
    unsigned history() const { return d_history; }

protected:
    void set_history(unsigned h) { d_history = h ? h : 1; }

private:
    unsigned d_history = 1;
};

}  // namespace jrc_host

namespace jrc_rt = jrc_host;
#define JRC_SPTR std::shared_ptr
#define JRC_GET_INITIAL_SPTR(p) JRC_SPTR<typename std::remove_pointer<decltype(p)>::type>(p)

#endif  // JRC_WITH_GNURADIO
