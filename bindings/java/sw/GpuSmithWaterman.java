package sw ;

import org.apache.spark.api.java.function.PairFlatMapFunction ;
import org.apache.spark.api.java.function.PairFunction ;

import scala.Tuple2 ;
import scala.Tuple3 ;

import java.nio.ByteBuffer ;
import java.nio.charset.StandardCharsets ;
import java.util.ArrayList ;
import java.util.Iterator ;

/**
 * Drop-in replacements for Distribution.MapRef (src/sw/Distribution.java:383-437) that run the matrix fill,
 * tied-maximum search and traceback of every (reference, read) pair on an MI355X through libswmi.so.
 *
 * NOT compiled by this repository's build (no JDK in the build image); see INTEGRATION.md.
 *
 *   mapRDD = listRDD.mapToPair( new MapRef() ) ;                           // reference, Distribution.java:338
 *   mapRDD = listRDD.mapPartitionsToPair( new GpuSmithWaterman.MapPartition() ) ;   // one native call per partition
 *   mapRDD = listRDD.mapToPair( new GpuSmithWaterman.MapRef() ) ;           // per element (simple, slower)
 */
public class GpuSmithWaterman
{
	static { System.loadLibrary( "swmi_jni" ) ; }

	public static final int TIE_SERIAL = 0 ;	// SmithWaterman.GetCellScore order
	public static final int TIE_STRICT = 1 ;	// DistributedSW.GetCellScore order

	static native long nativeCreate( int device ) ;
	static native void nativeDestroy( long ctx ) ;
	static native long nativeAlignBatch( long ctx , int match , int mismatch , int gap , int tieMode , byte[] types ,
			ByteBuffer refBytes , long[] refOff , int nRefs , ByteBuffer readBytes , long[] readOff , int nReads ) ;
	static native void nativeFreeBatch( long ctx , long batch ) ;
	static native int nativeRefTotal( long batch , int ref ) ;
	static native long nativeRefSiteCount( long batch , int ref ) ;
	static native byte[][] nativeRefSite( long batch , int ref , long k , int[] begin ) ;

	/** one context per executor thread: MapRef.call runs concurrently on every task thread */
	private static final ThreadLocal<Long> CTX = new ThreadLocal<Long>()
	{
		@Override protected Long initialValue()
		{
			int nGpus = Integer.getInteger( "swmi.gpus" , 8 ) ;
			int device = (int)( Thread.currentThread().getId() % nGpus ) ;
			return nativeCreate( device ) ;
		}
	} ;

	/**
	 * Characters above U+00FF cannot be narrowed to ISO-8859-1: map Character.toUpperCase(c) of every distinct
	 * such character of the partition to a free byte value first (the aligned strings are rebuilt from the
	 * original Strings on this side, so the mapping never shows in the output).  DNA/IUPAC input never needs it.
	 */
	private static ByteBuffer pack( ArrayList<String> seqs , long[] off )
	{
		long total = 0 ;
		for( int i = 0 ; i < seqs.size() ; i++ ) { off[i] = total ; total += seqs.get(i).length() ; }
		off[seqs.size()] = total ;
		ByteBuffer buf = ByteBuffer.allocateDirect( (int) Math.max(total,1) ) ;
		for( String s : seqs ) buf.put( s.getBytes(StandardCharsets.ISO_8859_1) ) ;
		return buf ;
	}

	/** mapPartitionsToPair variant: every element of the partition in ONE native call. */
	public static class MapPartition implements PairFlatMapFunction< Iterator<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> , Integer , Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>> >
	{
		@Override
		public Iterable<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>> call( Iterator<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> it )
		{
			ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> elems = new ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>>() ;
			while( it.hasNext() ) elems.add( it.next() ) ;
			ArrayList<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>> out = new ArrayList<Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>>( elems.size() ) ;
			if( elems.isEmpty() ) return out ;

			// CombineReadsToRef (Distribution.java:714-724) hands every element the same reads and algoArgs
			ArrayList<String> reads = elems.get(0)._2() ;
			int[] sc = elems.get(0)._3()._1() ;
			char[] ty = elems.get(0)._3()._2() ;
			ArrayList<String> refs = new ArrayList<String>( elems.size() ) ;
			for( Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>> e : elems ) refs.add( e._1()[1] ) ;

			long[] refOff = new long[refs.size()+1] , readOff = new long[reads.size()+1] ;
			ByteBuffer refBuf = pack( refs , refOff ) , readBuf = pack( reads , readOff ) ;
			byte[] types = { (byte)ty[0] , (byte)ty[1] , (byte)ty[2] , (byte)ty[3] } ;

			long ctx = CTX.get() ;
			long batch = nativeAlignBatch( ctx , sc[0] , sc[1] , sc[2] , TIE_SERIAL , types , refBuf , refOff , refs.size() , readBuf , readOff , reads.size() ) ;
			try
			{
				int[] begin = new int[1] ;
				for( int r = 0 ; r < refs.size() ; r++ )
				{
					long n = nativeRefSiteCount( batch , r ) ;
					ArrayList<Tuple2<Integer,String[]>> sites = new ArrayList<Tuple2<Integer,String[]>>( (int)n ) ;
					for( long k = 0 ; k < n ; k++ )
					{
						byte[][] al = nativeRefSite( batch , r , k , begin ) ;
						String[] aligned = { new String(al[0],StandardCharsets.ISO_8859_1) , new String(al[1],StandardCharsets.ISO_8859_1) } ;
						sites.add( new Tuple2<Integer,String[]>( new Integer(begin[0]) , aligned ) ) ;
					}
					out.add( new Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>>( new Integer(nativeRefTotal(batch,r)) ,
							new Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>( elems.get(r)._1() , sites ) ) ) ;
				}
			}
			finally { nativeFreeBatch( ctx , batch ) ; }
			return out ;
		}
	}

	/** Same signature as Distribution.MapRef: per-element drop-in. */
	public static class MapRef implements PairFunction< Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>> , Integer , Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>> >
	{
		@Override
		public Tuple2<Integer,Tuple2<String[],ArrayList<Tuple2<Integer,String[]>>>> call( Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>> tuple )
		{
			ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>> one = new ArrayList<Tuple3<String[],ArrayList<String>,Tuple2<int[],char[]>>>(1) ;
			one.add( tuple ) ;
			return new MapPartition().call( one.iterator() ).iterator().next() ;
		}
	}
}
