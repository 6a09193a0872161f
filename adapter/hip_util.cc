// adapter/hip_util.cc -- the Boda-side adapter of INTEGRATION.md section 1, kept as a source file so that it is COMPILED in this repository
// (tests/test_adapter_cpu.py builds it against adapter/shim/, which restates only the pieces of Boda's headers this file touches, links it with
// libbodahip.so and -- on a GPU -- drives Boda's own rtc_test flow through the C++ virtuals).  In a Boda checkout this
// file goes to src/hip_util.cc unchanged and is built by the stanza of INTEGRATION.md section 2 against the real headers.
// hip_util.cc -- be=hip: forwards rtc_compute_t to libbodahip.so (C ABI in bodahip.h)
#include"boda_tu_base.H"
#include"rtc_compute.H"
#include"str_util.H"
#include"bodahip.h"
namespace boda {
  static void hchk( int const rc ) {            // 0 ok | 1 unsupported -> unsup_err | 2 fatal -> rt_err
    if( rc == BODAHIP_UNSUPPORTED ) { unsup_err( bodahip_last_error() ); }
    if( rc != BODAHIP_OK ) { rt_err( bodahip_last_error() ); }
  }
  struct hdims_t {                               // dims_t -> bodahip_dims (borrowed pointers, lives for one call)
    vect_uint32_t sz; vector< char const * > nm; bodahip_dims d;
    hdims_t( dims_t const & x ) { for( uint32_t i = 0; i != x.sz(); ++i ) { sz.push_back( x.dims(i) ); nm.push_back( x.names(i).c_str() ); }
      d.tn = x.tn.c_str(); d.ndims = x.sz(); d.sizes = sz.empty() ? 0 : &sz[0]; d.names = nm.empty() ? 0 : &nm[0]; }
  };
  struct hip_compute_t : virtual public nesi, public rtc_compute_t // NESI(help="MI355X-native HIP/hiprtc rtc backend",
                         // bases=["rtc_compute_t"], type_id="hip" )
  {
    virtual cinfo_t const * get_cinfo( void ) const; // required declaration for NESI support
    uint32_t device; //NESI(default=0,help="HIP device ordinal (one process per GPU)")
    vect_uint32_t devices; //NESI(help="if non-empty: HIP device ordinals of ONE backend over several GPUs (vars with a leading img / M dim sharded, weights replicated)")
    bodahip_ctx * ctx;
    hip_compute_t( void ) : ctx(0) {}
    ~hip_compute_t( void ) { bodahip_destroy( ctx ); }
    void init( void ) { vector< int > ords( devices.begin(), devices.end() );
      hchk( ords.empty() ? bodahip_create( &ctx, device ) : bodahip_create_multi( &ctx, ords.size(), &ords[0] ) ); hchk( bodahip_set_gen_src( ctx, gen_src, gen_src_output_dir.exp.c_str() ) ); hchk( bodahip_init( ctx ) ); }
    string get_plat_tag( void ) { char b[512]; hchk( bodahip_get_plat_tag( ctx, b, sizeof(b) ) ); return b; }
    void create_var_with_dims( string const & vn, dims_t const & dims ) { hdims_t d(dims); hchk( bodahip_create_var( ctx, vn.c_str(), &d.d ) ); }
    void create_var_with_dims_as_reshaped_view_of_var( string const & vn, dims_t const & dims, string const & src_vn ) {
      hdims_t d(dims); hchk( bodahip_create_view( ctx, vn.c_str(), &d.d, src_vn.c_str() ) ); }
    void release_var( string const & vn ) { hchk( bodahip_release_var( ctx, vn.c_str() ) ); }
    dims_t get_var_dims( string const & vn ) {
      char tn[32], names[1024]; uint32_t nd = 16, sz[16]; hchk( bodahip_get_var_dims( ctx, vn.c_str(), tn, 32, &nd, sz, names, 1024 ) );
      dims_t r; r.tn = tn; char const * p = names; for( uint32_t i = 0; i != nd; ++i ) { r.add_dims( p, sz[i] ); p += strlen(p) + 1; } r.calc_strides(); return r; }
    void set_var_to_zero( string const & vn ) { hchk( bodahip_set_var_to_zero( ctx, vn.c_str() ) ); }
    void compile( vect_rtc_func_info_t const & fis, rtc_compile_opts_t const & o ) {
      vector< bodahip_func_info > hf; vector< vector< char const * > > an( fis.size() ); vect_string ops;
      for( uint32_t i = 0; i != fis.size(); ++i ) { ops.push_back( str( fis[i].op ) ); }   // NESI dump == the lexp line the ABI parses
      for( uint32_t i = 0; i != fis.size(); ++i ) { for( auto const & a : fis[i].arg_names ) { an[i].push_back( a.c_str() ); }
        hf.push_back( bodahip_func_info{ fis[i].func_name.c_str(), fis[i].func_src.c_str(), uint32_t(an[i].size()), an[i].empty() ? 0 : &an[i][0], ops[i].c_str() } ); }
      bodahip_compile_opts ho{ o.show_compile_log, o.enable_lineinfo, o.show_func_attrs, o.show_rtc_calls };
      hchk( bodahip_compile( ctx, hf.size(), hf.empty() ? 0 : &hf[0], &ho ) ); }
    void release_func( string const & fn ) { hchk( bodahip_release_func( ctx, fn.c_str() ) ); }
    void release_all_funcs( void ) { hchk( bodahip_release_all_funcs( ctx ) ); }
    uint32_t run( rtc_func_call_t const & rfc ) {
      vector< bodahip_arg > ha; std::list< hdims_t > keep;
      for( auto const & kv : rfc.arg_map ) { rtc_arg_t const & a = kv.second; assert_st( a.is_valid() );
        if( a.is_var() ) { ha.push_back( bodahip_arg{ kv.first.c_str(), 0, a.n.c_str(), {}, 0 } ); }
        else { keep.emplace_back( a.v->dims ); ha.push_back( bodahip_arg{ kv.first.c_str(), 1, 0, keep.back().d, a.v->rp_elems() } ); } }
      uint32_t id = 0; hchk( bodahip_run( ctx, rfc.rtc_func_name.c_str(), ha.size(), ha.empty() ? 0 : &ha[0], rfc.tpb.v, rfc.blks.v, &id ) ); return id; }
    void finish_and_sync( void ) { hchk( bodahip_finish_and_sync( ctx ) ); }
    void release_per_call_id_data( void ) { hchk( bodahip_release_per_call_id_data( ctx ) ); }
    float get_dur( uint32_t const & b, uint32_t const & e ) { float ms = 0; hchk( bodahip_get_dur( ctx, b, e, &ms ) ); return ms; }
    void profile_start( void ) { hchk( bodahip_profile_start( ctx ) ); }
    void profile_stop( void ) { hchk( bodahip_profile_stop( ctx ) ); }
    void copy_nda_to_var( string const & vn, p_nda_t const & nda ) { hdims_t d(nda->dims); hchk( bodahip_copy_to_var( ctx, vn.c_str(), &d.d, nda->rp_elems() ) ); }
    void copy_var_to_nda( p_nda_t const & nda, string const & vn ) { hdims_t d(nda->dims); hchk( bodahip_copy_from_var( ctx, nda->rp_elems(), &d.d, vn.c_str() ) ); }
    p_nda_t get_var_raw_native_pointer( string const & vn ) { void * p = 0; hchk( bodahip_get_raw_ptr( ctx, vn.c_str(), &p ) ); return make_shared<nda_t>( get_var_dims( vn ), p ); }
  };
#include"gen/hip_util.cc.nesi_gen.cc"
}
