// adapter/rtc_test_main.cc -- Boda's `rtc_test` mode (src/rtc_compute.cc:135-194) restated as a stand-alone program over the rtc_compute_t
// VIRTUALS of the be=hip adapter (adapter/hip_util.cc -> C ABI -> libbodahip.so): compile my_dot, three vars from vect_float, one launch,
// read back, check c == a + b.  With no argument it only constructs the backend object (link check; needs no GPU).
#include "boda_tu_base.H"
#include "rtc_compute.H"
#include <cmath>
#include <cstdio>
namespace boda {
p_rtc_compute_t make_rtc_compute_by_type_id( string const & be, uint32_t const device );
p_rtc_compute_t make_rtc_compute_by_type_id_devices( string const & be, vect_uint32_t const & devices );
static char const * const dot_src =
  "CUCL_GLOBAL_KERNEL void my_dot( GASQ float const * const a, GASQ float const * const b, GASQ float * const c, uint32_t const n ) {\n"
  "  uint32_t const ix = GLOB_ID_1D;\n  if( ix < n ) { c[ix] = a[ix] + b[ix]; }\n}\n";
int rtc_test_main( int argc, char ** argv ) {
  try {
    // "run": one device; "run-multi": the NESI field `devices` set to 0:0 -- ONE backend over two shards of GPU 0 (bodahip_create_multi)
    p_rtc_compute_t rtc = ( argc >= 2 && string( argv[1] ) == "run-multi" ) ? make_rtc_compute_by_type_id_devices( "hip", vect_uint32_t{ 0, 0 } ) : make_rtc_compute_by_type_id( "hip", 0 );
    if( argc < 2 ) { printf( "adapter linked: be=%s\n", rtc->be.c_str() ); return 0; }
    uint32_t const data_sz = 10000;
    rtc->init();
    op_base_t dot; dot.set_func_name( "my_dot" );
    rtc->compile( vect_rtc_func_info_t{ rtc_func_info_t{ dot.get_func_name(), dot_src, {"a","b","c","n"}, dot } }, rtc_compile_opts_t() );
    vect_float a( data_sz ), b( data_sz ), c( data_sz, 123.456f );
    for( uint32_t i = 0; i != data_sz; ++i ) { a[i] = 2.5f + 5.0f * float( ( i * 2654435761u ) >> 8 ) / 16777216.0f; b[i] = 7.5f - 5.0f * float( ( i * 40503u ) & 0xffff ) / 65536.0f; }
    rtc->init_var_from_vect_float( "a", a ); rtc->init_var_from_vect_float( "b", b ); rtc->init_var_from_vect_float( "c", c );
    uint32_t n = data_sz;
    rtc_func_call_t rfc; rfc.rtc_func_name = dot.get_func_name();
    rfc.arg_map["a"] = rtc_arg_t( "a" ); rfc.arg_map["b"] = rtc_arg_t( "b" ); rfc.arg_map["c"] = rtc_arg_t( "c" );
    rfc.arg_map["n"] = rtc_arg_t( make_shared<nda_t>( dims_t( {}, {}, "uint32_t" ), (void *)&n ) );
    rfc.tpb.v = 256; rfc.blks.v = ( data_sz + 255 ) / 256;
    uint32_t const id = rtc->run( rfc );
    rtc->finish_and_sync();
    float const ms = rtc->get_dur( id, id );
    rtc->set_vect_float_from_var( c, "c" );
    rtc->release_all_funcs();
    for( uint32_t i = 0; i != data_sz; ++i ) { if( std::fabs( ( a[i] + b[i] ) - c[i] ) > 1e-6f ) { printf( "bad res: i=%u a=%f b=%f c=%f\n", i, a[i], b[i], c[i] ); return 1; } }
    // an unsupported request must come back as unsup_err through the adapter (rc 1 of the C ABI)
    bool unsup = false;
    try { op_base_t bad; bad.set_func_name( "hip_sgemm" ); rtc->compile( vect_rtc_func_info_t{ rtc_func_info_t{ "sg", "", {"a","b","c"}, bad } }, rtc_compile_opts_t() );
      rtc->create_var_with_dims( "ha", dims_t( {4, 4}, {"K", "M"}, "double" ) ); rtc->create_var_with_dims( "hb", dims_t( {4, 4}, {"K", "N"}, "double" ) ); rtc->create_var_with_dims( "hc", dims_t( {4, 4}, {"M", "N"}, "double" ) );
      rtc_func_call_t r2; r2.rtc_func_name = "sg"; r2.arg_map["a"] = rtc_arg_t( "ha" ); r2.arg_map["b"] = rtc_arg_t( "hb" ); r2.arg_map["c"] = rtc_arg_t( "hc" ); rtc->run( r2 );
    } catch( unsup_exception const & ) { unsup = true; }
    if( !unsup ) { printf( "double-typed hip_sgemm did not raise unsup_err through the adapter\n" ); return 1; }
    printf( "All is Well. plat_tag=%s my_dot %.4f ms\n", rtc->get_plat_tag().c_str(), ms );
    return 0;
  } catch( std::exception const & e ) { printf( "error: %s\n", e.what() ); return 2; }
}
} // namespace boda
int main( int argc, char ** argv ) { return boda::rtc_test_main( argc, argv ); }
